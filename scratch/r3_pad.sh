#!/bin/bash
# resident problems per CU against time: does the spill arena falling out of L2 cost time?  (LMPC_LDS_PAD, measurement only)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_facade.py -q -x -k "step_for_step" -s 2>&1 | tail -8 > gpurun_out/node_step.log
for pad in 0 3200 7100 12500; do
  echo "== LMPC_LDS_PAD=$pad" >> gpurun_out/pad.log
  LMPC_LDS_PAD=$pad python scratch/r3_time.py trk20 2>&1 | grep "polish=on" >> gpurun_out/pad.log
done
for pad in 0 2200 6100 11500; do
  echo "== lmpc LMPC_LDS_PAD=$pad" >> gpurun_out/pad.log
  LMPC_LDS_PAD=$pad python scratch/r3_time.py lmpc 2>&1 | grep "B= 32768" | grep "polish=on :" >> gpurun_out/pad.log
done
cat gpurun_out/node_step.log gpurun_out/pad.log
