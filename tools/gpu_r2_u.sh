#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_clk
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d gpurun_out/prof_clk -o clk -- python tools/scan_bench.py --reps 2 --only-scan > gpurun_out/prof_clk.log 2>&1
python - <<'PY'
import csv,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open('gpurun_out/prof_clk/clk_counter_collection.csv')):
    k=r['Kernel_Name']
    for n in ('scan_fwd_kernel','scan_bwd_kernel','reduce_partials'):
        if n in k:
            key=n+'_g'+r['Grid_Size']
            acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
            acc[key]['ns'].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,d in sorted(acc.items()):
    m={c:sum(v)/len(v) for c,v in d.items()}
    print(k, 'ns=%.0f'%m['ns'], 'GUI/8=%.0f'%(m['GRBM_GUI_ACTIVE']/8), 'clock_GHz=%.3f'%(m['GRBM_GUI_ACTIVE']/8/m['ns']), {c:'%.3g'%v for c,v in m.items() if c.startswith('SQ')})
PY
