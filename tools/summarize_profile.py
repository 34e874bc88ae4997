"""Turn a rocprofv3 `--kernel-trace --stats --output-format csv` run into the markdown summary kept
under profiles/.

    python tools/summarize_profile.py gpurun_out/prof_r01e profiles/r01 \
        --bench-line gpurun_out/r01e_bench_line.json --command "<the profiled command>"

Copies <dir>/**/*kernel_stats.csv to <prefix>_bench_kernel_stats.csv, the bench line to
<prefix>_bench_line.json, and writes <prefix>_summary.md.
"""
import argparse
import csv
import glob
import json
import os
import shutil

FLOP_PER_SAMPLE_CNL = 923136
PEAK = 157.3e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('prof_dir')
    ap.add_argument('prefix')
    ap.add_argument('--bench-line')
    ap.add_argument('--command', default='')
    ap.add_argument('--round', default='1')
    args = ap.parse_args()

    stats = sorted(glob.glob(os.path.join(args.prof_dir, '**', '*kernel_stats.csv'), recursive=True))
    assert stats, f'no *kernel_stats.csv under {args.prof_dir}'
    dst_csv = args.prefix + '_bench_kernel_stats.csv'
    shutil.copyfile(stats[0], dst_csv)
    rows = list(csv.DictReader(open(dst_csv)))
    line = None
    if args.bench_line:
        txt = [l for l in open(args.bench_line).read().splitlines() if l.startswith('{')]
        line = json.loads(txt[-1])
        json.dump(line, open(args.prefix + '_bench_line.json', 'w'), indent=1)

    out = [f'# Round {args.round} -- rocprofv3 kernel summary of the headline bench', '']
    if args.command:
        out += [f'Command (on the MI355X box): `{args.command}`', '']
    out += [f'Full table: `{os.path.basename(dst_csv)}`; the bench line of the un-profiled run: '
            f'`{os.path.basename(args.prefix)}_bench_line.json`.', '',
            '| kernel | calls | avg ms | min ms | max ms | % of GPU time |', '|---|---|---|---|---|---|']
    mlp = None
    for r in rows[:14]:
        name = r['Name']
        out.append(f"| `{name[:64]}` | {r['Calls']} | {float(r['AverageNs']) / 1e6:.3f} | "
                   f"{float(r['MinNs']) / 1e6:.3f} | {float(r['MaxNs']) / 1e6:.3f} | {float(r['Percentage']):.2f} |")
        if name.startswith('occ::m16::canonical_mlp_lds_kernel'):
            mlp = r
    out.append('')
    if mlp is not None and line is not None:
        n = int(line['config'].get('samples_evaluated_per_frame',
                                   line['config']['rays_per_frame'] * line['config']['samples_per_ray']))
        avg = float(mlp['AverageNs']) * 1e-9
        trace = sorted(glob.glob(os.path.join(args.prof_dir, '**', '*kernel_trace.csv'), recursive=True))
        if trace:                       # the timed region = the launches after the warm-up steps
            d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9 for r in csv.DictReader(open(trace[0]))
                 if r['Kernel_Name'].startswith('occ::m16::canonical_mlp_lds_kernel')]
            timed = d[line['warmup']:]
            out += ['`canonical_mlp_lds_kernel` launches in order (ms): ' + ', '.join(f'{x * 1e3:.2f}' for x in d)
                    + f" -- the first {line['warmup']} are warm-up steps; timed-region average "
                    f'{sum(timed) / len(timed) * 1e3:.2f} ms.', '']
            avg = sum(timed) / len(timed)
        tf = FLOP_PER_SAMPLE_CNL * n / avg
        rl = line['roofline']
        out += [f"`canonical_mlp_lds_kernel`: {avg * 1e3:.2f} ms timed-region average for {n} samples x {FLOP_PER_SAMPLE_CNL} FLOP = "
                f"{FLOP_PER_SAMPLE_CNL * n / 1e12:.2f} TFLOP -> {tf / 1e12:.1f} TFLOP/s = {100 * tf / PEAK:.1f} % of the "
                f"157.3 TFLOP/s fp32-MFMA peak. bench.py's HIP-event measurement of the same launches in the "
                f"un-profiled run: {rl['launch_ms']:.2f} ms ({100 * abs(rl['launch_ms'] - avg * 1e3) / (avg * 1e3):.2f} % "
                f"apart), frac {rl['frac']:.3f}.", '']
        out += [f"Bench line: {line['value']:.0f} rays/s, {line['ms_per_step']:.1f} ms/frame"
                + (f"; opt-in bf16x3 path {line['alt']['value']:.0f} rays/s, {line['alt']['ms_per_step']:.1f} ms/frame"
                   if 'alt' in line else '')
                + (f"; every sample evaluated {line['all_samples']['value']:.0f} rays/s, {line['all_samples']['ms_per_step']:.1f} ms/frame"
                   if 'all_samples' in line else '')
                + (f"; CPU oracle {line['cpu_baseline']['value']:.0f} rays/s on {line['cpu_baseline']['cores']} cores"
                   if 'cpu_baseline' in line else '') + '.', '']
    open(args.prefix + '_summary.md', 'w').write('\n'.join(out))
    print('\n'.join(out))


if __name__ == '__main__':
    main()
