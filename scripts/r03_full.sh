#!/bin/bash
# the whole -m gpu suite + smoke(), as the driver runs them at round end
mkdir -p gpurun_out/r03_full
python -m pytest tests -x -q -m gpu > gpurun_out/r03_full/pytest.log 2>&1; rc=$?
tail -n 15 gpurun_out/r03_full/pytest.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 3
