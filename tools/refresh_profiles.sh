#!/bin/bash
# Usage (GPU box, repo root): tools/refresh_profiles.sh <tag>   -> gpurun_out/<tag>/...
# The un-profiled bench line, the rocprofv3 kernel summary + trace of the headline run and of a frame with the renderer's
# default (repeated samples eliminated), and the MFMA / HBM counter passes (each --pmc pass on its own, with a time limit).
tag=$1
export TMPDIR=/tmp
o=gpurun_out/$tag
mkdir -p $o
timeout -k 5 900 python3 bench.py > $o/bench_line.json 2> $o/bench.err
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_bench -o b -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt > $o/prof_bench_line.json 2> $o/prof_bench.err
cat > /tmp/dd_frame.py <<'PY'
import sys, torch
sys.path.insert(0, '.')
from occnerf_amd import synth
from occnerf_amd.seeded import build_network, frame_to_device
torch.set_grad_enabled(False)
net = build_network(0, False, S=128, non_rigid=True)
frame = synth.make_frame(img_size=512, pose72=synth.seeded_pose(1), orbit_frame=28)
data = frame_to_device(frame, 'cuda:0')
for _ in range(6):
    net(**data, iter_val=1e7, ray_order_key='k')
torch.cuda.synchronize()
print(int(net.last_live_count), [int(x) for x in net.last_head_counts])
PY
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_dedup -o d -- python3 /tmp/dd_frame.py > $o/prof_dedup.log 2>&1
timeout -k 5 400 bash tools/pmc_mfma.sh $o/pmc_mfma.json > $o/pmc_mfma.log 2>&1
timeout -k 5 300 bash tools/pmc_hbm.sh $o/pmc_hbm.json > $o/pmc_hbm.log 2>&1
# calibration of FETCH_SIZE for streams vs gathers (tools/fetch_calib.hip), and the MFMA issue-rate microbenchmark
hipcc --offload-arch=gfx950 -O3 -o /tmp/fc tools/fetch_calib.hip 2>/dev/null
for m in 0 1 2; do
  timeout -k 5 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/fc$m -o p -- /tmp/fc $m > $o/fc$m.log 2>&1
done
hipcc --offload-arch=gfx950 -O3 -o /tmp/mir tools/mfma_issue_rate.hip 2>/dev/null && /tmp/mir > $o/mfma_issue_rate.txt 2>&1
# the split-operand (f16x3 / bf16x3) MLP kernels: MFMA counters of the alt2 leg
timeout -k 5 400 bash tools/pmc_mfma.sh $o/pmc_mfma_alt2.json "--only alt2" > $o/pmc_mfma_alt2.log 2>&1
# the training step: kernel summary, torch-level profile, HBM counters of every kernel
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_train -o t -- python3 tools/train_step_trace.py 10 > $o/train_trace.log 2>&1
timeout -k 5 300 python3 tools/train_step_profile.py > $o/train_step_profile_bf16.txt 2>&1
timeout -k 5 400 bash tools/pmc_train.sh $o/pmc_train.json > $o/pmc_train.log 2>&1
ls $o
