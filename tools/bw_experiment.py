"""Experiments on the BigWig-files-to-result leg: the files are written once (WTAMD_BENCH_BWDIR), then every
configuration (a set of environment variables) runs in its own process.  Usage:
    python tools/bw_experiment.py <mbp> 'NAME=V NAME2=V2' 'NAME=V' ...     ('' = defaults)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mbp = sys.argv[1]
env0 = dict(os.environ, WTAMD_BENCH_BWDIR="/dev/shm/wtamd_exp", WTAMD_BENCH_NO_HOSTDEC="1")
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e_bw_only.py"), mbp], env=env0, capture_output=True, text=True)   # writes the files
for cfg in sys.argv[2:]:
    env = dict(env0)
    for kv in cfg.split():
        k, v = kv.split("=", 1)
        env[k] = v
    env["WTAMD_E2E_REPS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e_bw_only.py"), mbp], env=env, capture_output=True, text=True)
    for line in r.stdout.strip().splitlines():
        try:
            d0 = json.loads(line)
            for kind in ("cold", "warm"):
              d = d0[kind]
              print(kind, "%-60s total %.3f s  %.3e bp/s  steady %.3e  open %.3f (readers %.3f)  submit %.0f ms  wait %.0f ms  decode %.0f ms  batches %d"
                  % (cfg or "(defaults)", d["seconds"], d["bp_per_s"], d.get("steady_bp_per_s", 0), d["open_seconds"], d["open_readers_seconds"],
                     d["host_submit_ms"], d["host_wait_ms"], d["sum_device_decode_ms"], d["batches"]), flush=True)
        except Exception as e:
            print(cfg, "FAILED", repr(e), r.stderr[-400:], flush=True)
