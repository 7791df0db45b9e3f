cd $GRAFT_REPO_ROOT
TESTS=1 TESTS_K="difference_array or exact or golden or config_sized or plans_agree or full_size" tools/r6_ab.sh ab1 base - "c2 20 16" "c2 0 16" "c2 20 200"
tools/r6_ab.sh ab1 dpp - "c2 0 16"
