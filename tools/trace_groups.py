"""Group a rocprofv3 kernel trace by (kernel, workgroups): usage trace_groups.py DIR STEPS [pattern] [topn]"""
import csv, collections, glob, sys
d, steps = sys.argv[1], float(sys.argv[2])
pat = sys.argv[3] if len(sys.argv) > 3 else ''
topn = int(sys.argv[4]) if len(sys.argv) > 4 else 30
f = (glob.glob(d + '/*kernel_trace.csv') + glob.glob(d + '/*/*kernel_trace.csv'))[0]
agg = collections.defaultdict(lambda: [0, 0.0])
trace = list(csv.DictReader(open(f)))
if steps <= 0:      # 0 = count the steps: the NoiseLayer kernel runs once per f_step
    steps = float(sum(1 for r in trace if 'nast_noise_kernel' in r['Kernel_Name']) or 1)
for r in trace:
    if pat not in r['Kernel_Name']: continue
    name = r['Kernel_Name'].split('(')[0][-34:]
    key = (name, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']))
    agg[key][0] += 1; agg[key][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"total {tot/steps/1e3:.2f} ms/step")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:topn]:
    print(f"{k[0]:36s} wgs={k[1]:6d} n/step={v[0]/steps:6.1f} avg_us={v[1]/v[0]:9.1f} ms/step={v[1]/steps/1e3:8.2f} {100*v[1]/tot:5.1f}%")
