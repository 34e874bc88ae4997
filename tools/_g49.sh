export TMPDIR=/tmp
o=gpurun_out/r05v
mkdir -p $o
python3 bench.py > $o/bench_line.json 2> $o/bench.err
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_train -o t -- python3 tools/train_step_trace.py 10 > $o/train_trace.log 2>&1
python3 tools/train_step_profile.py > $o/train_step_profile_bf16.txt 2>&1
tools/pmc_train.sh $o/pmc_train.json > $o/pmc_train.log 2>&1
python3 - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r05v/bench_line.json') if l.startswith('{')][-1])
print('headline', j['value'], j['ms_per_step'], j['median_ms_per_step'], j['roofline']['frac'], j['roofline']['end_to_end_frac'], j['roofline']['launch_ms'])
for k in ('dedup','alt','alt2','no_shortcuts','all_samples','rccl_world1'):
    v=j.get(k); print(k, round(v['ms_per_step'],2), round(v['value']), (round(v['default']['ms_per_step'],2), round(v['default']['value'])) if 'default' in v else '')
print('alt2 mlp', j['alt2'].get('canonical_mlp_launch_ms'))
for k in ('movement','config4'):
    v=j[k]; print(k, round(v['every_live_sample']['ms_per_frame'],1), round(v['every_live_sample']['value']), round(v['default']['ms_per_frame'],1), round(v['default']['value']))
v=j['freeview_orbit']; print('orbit', v['ms_per_frame'], v['value'], v['ray_order_ms_per_frame']['hip_kernels_gpu'], v['ray_order_ms_per_frame']['torch_ops_gpu'])
t=j['train']; print('train', t['ms_per_step'], t['scattered_ms_per_step'], t['roofline']['frac'], t['roofline']['traffic'])
print({k:(round(v['slowest_rank_ms'],2), round(v['T1_over_slowest'],3), round(v['per_rank_fixed_ms'],2)) for k,v in j['predicted_scaling']['worlds'].items()}, j['predicted_scaling']['T1_ms'])
print('cpu', j['cpu_baseline']['value'], 'nr', j['roofline']['nonrigid'])
PY
tail -3 $o/pmc_train.log
