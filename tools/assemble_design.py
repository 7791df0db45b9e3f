"""One-off (round 5): DESIGN.md rewritten as a CURRENT-STATE document.  The old file's sections (cut into /tmp/design_parts by
heading) are put in numeric order, kernel by kernel, with cross references renumbered; what is history (per-round diaries,
"tried and dropped" lists, review tables) goes to the appendix; the sections written new come from tools/design_new/*.md.
Usage: git show f894be8:DESIGN.md | python tools/split_design_r4.py /tmp/design_parts; python tools/assemble_design.py /tmp/design_parts tools/design_new > /tmp/D.md;
       python tools/fill_design.py /tmp/D.md > DESIGN.md"""
import glob
import os
import re
import sys

parts_dir, new_dir = sys.argv[1], sys.argv[2]


def part(n):
    f = glob.glob(os.path.join(parts_dir, "%03d_*.md" % n))[0]
    return open(f).read().rstrip("\n")


def new(name):
    return open(os.path.join(new_dir, name + ".md")).read().rstrip("\n")


# old section number -> new one
MAP = {"4.1": "4.1", "4.2": "4.3", "4.3": "4.7", "4.4": "4.8", "4.5": "4.8", "4.6": "4.10", "4.7": "4.4", "4.8": "4.9", "4.9": "4.2",
       "4.10": "4.8", "4.11": "4.5", "5": "5", "6": "8", "7": "9", "8": "10", "9": "11", "10": "A.1", "11": "6", "11.1": "6.1",
       "11.2": "6.2", "11.3": "6.3", "11.4": "6.8", "11.5": "6.4", "12": "A.2", "13": "6.5", "13.1": "A.5", "13.1b": "6.6", "13.2": "A.5",
       "13.3": "4.9", "13.4": "6.6", "13.5": "6.7", "13.6": "6.8", "14": "7", "15": "8", "16": "A.3", "16.1": "A.3", "17": "A.4",
       "17.1": "A.4", "1": "1", "2": "2", "3": "3", "4": "4"}


def renumber(text):
    def one(m):
        old = m.group(2)
        return m.group(1) + MAP.get(old, old)
    text = re.sub(r"(§)(\d+(?:\.\d+b?)?)", one, text)
    text = re.sub(r"(DESIGN(?:\.md)? )(\d+(?:\.\d+b?)?)(?![\d:])", one, text)
    return text


def body(n, heading=None):
    """old part n with its heading replaced (None: heading dropped)"""
    t = part(n).split("\n")
    t = t[1:]
    while t and not t[0].strip():
        t = t[1:]
    t = renumber("\n".join(t))
    return (heading + "\n\n" + t) if heading else t


out = []
add = out.append
add("# DESIGN — MI355X-native multiplexer / reducer engine for WiggleTools")
add("")
add("Reference: Ensembl/WiggleTools v1.2.11 (`/root/reference`, cited as `path:line`).  Scope contract: `SURVEY.md` §8.")
add("This file is the CURRENT state (round 5): the path and its boundary, the data layout in HBM, every kernel with the roofline")
add("that bounds it and today's measurement, what is out of scope.  How each kernel got where it is -- what was tried, round by")
add("round -- is Appendix A; the numbers of earlier rounds are Appendix B.")
add("")
add(new("00_state"))
add("")
add(body(0, "## 1. The path and its boundary").replace(
    "* round 3: device-side BigWig decode", "History in one paragraph each (details: Appendix A).\n\n* round 3: device-side BigWig decode"))
add("")
add(new("01_round5_bullets"))
add("")
add(body(1, "## 2. Oracle and what pins it"))
add("")
add(body(2, "## 3. Data layout in HBM"))
add("")
add(new("04_kernels_intro"))
add("")
add(new("041_delta_today"))
add("")
_b4 = body(4, "#### How the kernel works")
_i0 = _b4.index("What bounded it, step by step")
_i1 = _b4.index("**A hipcc (ROCm 7.2) hazard")
_bounded = _b4[_i0:_i1].rstrip()
add(_b4[:_i0] + _b4[_i1:])
add("")
add(body(12, "### 4.2 `wt_delta_kernel` with squares — Variance / StdDev / Entropy / CV (`QQ` templates, `csrc/wt_delta.h`)"))
add("")
add(new("042_squares_today"))
add("")
add(body(5, "### 4.3 `wt_reduce_kernel<OP, ValT, ScrT, K, MULTI>` — the fused bitmap multiplexer + reducer"))
add("")
add(body(9, "### 4.4 Median / MWU in the bitmap kernel: the value column in registers (`wt_gather_regs`, `wt_sort_regs`, `wt_mwu_regs`)"))
add("")
add(new("044_erf_table"))
add("")
add(body(10, "### 4.5 MedianReduction by walking: `wt_walk_kernel` (`csrc/wt_walk.h`, `csrc/wt_walk.hip`)"))
add("")
add(new("045_walk_round5"))
add("")
add(new("046_mwalk"))
add("")
add(body(6, "### 4.7 Window index: `wt_index_coarse_kernel` + `wt_index_search_kernel` (`wt_index_kernel` = the scan)"))
add("")
add("### 4.8 Integrators, run compression, the `map`-able operators")
add("")
add(body(7, "#### `wt_auc_kernel`, `wt_pearson_kernel`, `wt_extents_kernel`, `wt_compress_*`"))
add("")
add(body(8, "#### `wm_map_kernel` / `wm_compact_kernel` — the `map`-able unary operators (`csrc/wt_map.hip`)"))
add("")
add(body(13, "#### Operator chains inside the pipeline (`wt_map_chain_async`, `wtamd_pipe_set_map`)"))
add("")
add("### 4.9 Pipeline kernels and the BigWig kernels")
add("")
add(body(11, "#### `wt_gather_kernel`, `wt_export_kernel` (`csrc/wt_pipe.h`)"))
add("")
add(body(28, "#### BigWig sections decoded on the device (`csrc/wt_inflate.h`, `csrc/wt_bwdev_core.h`, `csrc/wt_bwdev.hip`)"))
add("")
add(body(32, "#### The inflate kernel as it stands (rewritten in round 4)"))
add("")
add(new("049_inflate_round5"))
add("")
add(body(14, "### 4.10 Roofline accounting (bound: HBM)"))
add("")
add(new("05_measurement"))
add("")
add("## 6. The streaming pipeline and the files (north-star N1; `csrc/wt_pipe.h`, `csrc/wt_iter_abi.cpp` + `csrc/wt_abi_*.h`)")
add("")
add(body(21))
add("")
add(body(22, "### 6.1 Slots and streams"))
add("")
add(body(23, "### 6.2 Feeding it from lazy iterators"))
add("")
add(body(24, "### 6.3 Bulk doors"))
add("")
add(body(25, "### 6.4 Files: `wtamd_BigWiggleReader`"))
add("")
add(new("065_device_files"))
add("")
add(new("066_cold_start"))
add("")
add(body(30, "#### Pools (round 3)"))
add("")
add(body(33, "#### Memory of a file-byte pipe (round 4)"))
add("")
add(body(34, "### 6.7 When the device decoder says no"))
add("")
add(new("068_measured"))
add("")
add(body(36, "## 7. Fused integrators through the reference API"))
add("")
add(body(16, "## 8. Multi-GPU"))
add("")
add(body(37, "### 8.1 One pipeline per GPU inside the drop-in layer"))
add("")
add(new("082_rccl_world1"))
add("")
add(body(17, "## 9. Out of scope (and why)"))
add("")
add(body(18, "## 10. Test infrastructure that is not product"))
add("")
add(new("11_coverage"))
add("")
add(new("12_round5_review"))
add("")
add("## Appendix A. How the kernels got here (history, round by round)")
add("")
add("Kept for the reader who wants to know what was tried before trying it again.  Section numbers inside these texts have been")
add("mapped to the current ones.")
add("")
add("### A.1 Where the time went (rounds 1-3; `-DWT_PROFILE` builds, SQ counters)")
add("")
add("`wt_delta_kernel`, from the first version to round 3: " + _bounded)
add("")
add(body(20))
add("")
add(body(27, "### A.2 Round 2 against the round-1 review"))
add("")
add(body(38, "### A.3 Round 3 against the round-2 review"))
add("")
add(body(39, "#### Leads of round 3 (measured then, acted on in round 4 unless noted)"))
add("")
add(body(40, "### A.4 Round 4 against the round-3 review"))
add("")
add(body(41, "#### Leads of round 4 (what round 5 did with them: §12)"))
add("")
add(body(29, "### A.5 The file leg, step by step (round 3: chr1 x 100 files = 1.52·10⁹ intervals, 7.28 GB of files)"))
add("")
add(body(31, "#### Not done in round 3"))
add("")
add(body(35, "#### Whole-genome files (round 4's record)"))
add("")
add(new("B_numbers_by_round"))
add("")
add(body(15, "### B.1 Measurement text and tables of rounds 2-4 (as written then)"))
print("\n".join(out))
