#!/bin/bash
# Usage (GPU box, repo root): tools/pmc_mfma.sh <out.json> [bench legs, default --no-alt; e.g. "--only alt2" adds the f16x3 kernels]
# MFMA utilisation of the two matrix-pipe kernels of the bench frame (north star: "rocprof ... MFMA utilisation against
# chip peak").  Three separate --pmc passes, the program directly after `--` (never combined with trace domains):
#   SQ_INSTS_MFMA, SQ_VALU_MFMA_BUSY_CYCLES | SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_INSTS_VALU | GRBM_GUI_ACTIVE
out=$1
export TMPDIR=/tmp
d=gpurun_out/pmc_mfma
mkdir -p $d
legs=${2:---no-alt}
args="bench.py --steps 2 --warmup 1 --no-cpu-baseline $legs"
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $d/p1 -o p -- python3 $args > $d/p1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $d/p2 -o p -- python3 $args > $d/p2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $d/p3 -o p -- python3 $args > $d/p3.log 2>&1
python3 - $d $out <<'PY'
import csv, glob, json, sys, collections
d, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(f'{d}/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if any(t in k for t in ('canonical_mlp_lds_kernel', 'nonrigid_lds_kernel', 'canonical_mlp_split_lds_kernel', 'canonical_mlp_split_tail_kernel',
                                'nonrigid_split_kernel')):
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
            if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
                dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9)
line = [l for l in open(f'{d}/p1.log').read().splitlines() if l.startswith('{')]
n = int(json.loads(line[-1])['config']['samples_evaluated_per_launch']) if line else None
SIMDS = 256 * 4
XCDS = 8            # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (2.14e9 for a 113 ms launch = 8 x 2.37 GHz)
res = {'source': 'tools/pmc_mfma.sh: three rocprofv3 --pmc passes (SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES | SQ_BUSY_CYCLES '
                 'SQ_WAVE_CYCLES SQ_INSTS_VALU | GRBM_GUI_ACTIVE) of `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline '
                 '--no-alt` on MI355X; means over the launches',
       'notes': 'SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs (MI355X_MICROARCH.md: = issue cycles x N_mfma); '
                'mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); one v_mfma_f32_16x16x4_f32 '
                'occupies its SIMD for 32 cycles, so mfma_issue_frac = SQ_INSTS_MFMA x 32 / (GRBM_GUI_ACTIVE / 8 x 1024); '
                'profiled passes run at a lower clock than un-profiled ones (DVFS), ratios are unaffected',
       'samples_per_launch': n, 'kernels': {}}
for k, c in acc.items():
    m = {name: sum(v) / len(v) for name, v in c.items()}
    g = m.get('GRBM_GUI_ACTIVE')
    t = sum(dur[k]) / len(dur[k]) if dur[k] else None
    e = dict(m)
    if g:
        cyc = g / XCDS                                    # busy cycles of the launch on one XCD's clock
        e['mfma_busy_frac'] = m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cyc * SIMDS)
        e['mfma_issue_frac'] = m.get('SQ_INSTS_MFMA', 0) * 32 / (cyc * SIMDS)
        e['launch_ms_profiled'] = t * 1e3
        e['clock_ghz_profiled'] = cyc / t / 1e9
        # (the split kernels issue v_mfma_f32_32x32x16_{f16,bf16}: 32 768 flops and the same 32 cycles per instruction)
        per = 2 * 32 * 32 * 16 if 'split' in k else 2 * 16 * 16 * 4
        e['mfma_tflops_executed'] = m.get('SQ_INSTS_MFMA', 0) * per / t / 1e12
    e['launches'] = len(c.get('SQ_INSTS_MFMA', []))
    res['kernels'][k] = e
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res['kernels'], indent=1))
PY
