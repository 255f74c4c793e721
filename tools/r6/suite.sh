#!/bin/bash
# the GPU suite on the current sources with their stamp + smoke() -> gpurun_out/$1/gpu_suite.txt
R=${GRAFT_REPO_ROOT:?run through gpurun}
O=$R/gpurun_out/${1:-r6_suite}
mkdir -p $O
cd $R
SHA=$(python -c "import bench; print(bench.csrc_sha16())")
( echo "# pytest tests -x -q -m gpu on MI355X; csrc_sha16 = $SHA; git-tracked sources as of the commit that holds this file"; python -m pytest tests -x -q -m gpu --durations=8; echo "# __graft_entry__.smoke():"; python -c "import __graft_entry__ as g; g.smoke()" ) > $O/gpu_suite.txt 2>&1
tail -6 $O/gpu_suite.txt
