#!/bin/bash
mkdir -p gpurun_out/r05
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > gpurun_out/r05/smoke.log 2>&1
tail -3 gpurun_out/r05/smoke.log
