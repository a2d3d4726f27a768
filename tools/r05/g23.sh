#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05/g23; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -o int -- python3 tools/timeline_int.py > $O/run.txt 2>&1
tail -3 $O/run.txt
find $O/trace -name '*.csv' | head; 
for f in $(find $O/trace -name '*kernel_trace.csv'); do python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), rows[0].keys())
PY
done
# keep only a compact version: name,start,end
for f in $(find $O/trace -name '*_trace.csv'); do
python3 - "$f" $O <<'PY'
import csv, sys, os
f, O = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(f)))
kind = 'kernel' if 'kernel' in os.path.basename(f) else 'copy'
with open(os.path.join(O, kind + '_compact.csv'), 'w') as out:
    for r in rows:
        if kind == 'kernel':
            out.write(f"{r['Start_Timestamp']},{r['End_Timestamp']},{r['Kernel_Name'][:90].replace(',', ';')}\n")
        else:
            out.write(f"{r['Start_Timestamp']},{r['End_Timestamp']},{r.get('Direction', '')},{r.get('Bytes', r.get('Size', ''))}\n")
PY
done
rm -rf $O/trace; ls -la $O
