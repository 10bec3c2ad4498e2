"""Summarise a rocprofv3 --kernel-trace --stats (csv) output dir: per-kernel totals per step.
usage: prof_summary.py DIR STEPS [TOPN]     (STEPS = 0: the number of nast_noise_kernel launches = f_steps in the run)"""
import csv, glob, sys
d, steps = sys.argv[1], float(sys.argv[2])
f = (glob.glob(d + '/*kernel_stats.csv') + glob.glob(d + '/*/*kernel_stats.csv'))[0]
rows = list(csv.DictReader(open(f)))
if steps <= 0:      # 0 = count the steps: the NoiseLayer kernel runs once per f_step
    steps = float(sum(int(r['Calls']) for r in rows if 'nast_noise_kernel' in r['Name']) or 1)
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"kernel time per step: {tot/1e6/steps:.2f} ms ; launches per step: {sum(int(r['Calls']) for r in rows)/steps:.0f}")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    print(f"{r['Name'][:84]:84s} calls/step={int(r['Calls'])/steps:7.1f} ms/step={float(r['TotalDurationNs'])/1e6/steps:7.3f} avg_us={float(r['AverageNs'])/1e3:8.2f} {float(r['Percentage']):5.1f}%")
