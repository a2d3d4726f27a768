#!/bin/bash
# clock during the in-step and the replayed launches: GRBM_GUI_ACTIVE (cycles) over the kernel-trace duration, per dispatch
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03_gap2; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES -d $O/p -o p --output-format csv -- python3 tools/gap_probe.py > $O/gap.md 2>&1
python3 - <<'PY'
import csv, glob, collections
base = 'gpurun_out/r03_gap2/p/'
cc = glob.glob(base + '**/*counter_collection.csv', recursive=True)
kt = glob.glob(base + '**/*kernel_trace.csv', recursive=True)
print(cc, kt)
dur = {}
for r in csv.DictReader(open(kt[0])):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp']), int(r['Start_Timestamp']), r['Kernel_Name'])
vals = collections.defaultdict(dict)
for r in csv.DictReader(open(cc[0])):
    vals[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
rows = []
for d, (ns, t0, name) in dur.items():
    if 'k_conv_wave' not in name or d not in vals or ns < 300000: continue
    v = vals[d]
    rows.append((t0, int(d), ns / 1e3, v.get('GRBM_GUI_ACTIVE', 0), v.get('SQ_BUSY_CYCLES', 0), v.get('SQ_WAVES', 0), name[:40]))
rows.sort()
with open('gpurun_out/r03_gap2/clock.txt', 'w') as f:
    for t0, d, us, gui, busy, waves, name in rows:
        f.write(f'{d:7d} t={t0 / 1e6:12.3f} ms  {us:8.1f} us  GUI {gui:12.0f}  -> {gui / us:8.1f} cycles/us  SQ_BUSY {busy:12.0f} waves {waves:8.0f} {name}\n')
PY
head -100 $O/clock.txt
