#!/bin/bash
# SQ counters of the kernels of the default training step (rocprofv3 --pmc passes over a short bench run), per-dispatch averages per kernel.
# usage (inside gpurun): tools/pmc_step.sh <tag> [kernel-name regex, default: conv_]
tag=$1; pat=${2:-conv_}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmcs_${tag}_$i -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 3 --no-prof --no-cpu-baseline --no-config5 > $R/gpurun_out/pmcs_${tag}_$i.log 2>&1 || { tail -5 $R/gpurun_out/pmcs_${tag}_$i.log; }
done
python3 - <<PY
import csv, glob, collections, re, json
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob('$R/gpurun_out/pmcs_${tag}_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if not re.search(r'$pat', k): continue
        k = re.sub(r'\(.*', '', k.replace('dsrl::', '').replace('void ', ''))
        a = agg[k][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
out = {}
for k, cs in agg.items():
    d = {c: v[0] / v[1] for c, v in cs.items()}
    d['dispatches'] = max(v[1] for v in cs.values())
    if 'GRBM_GUI_ACTIVE' in d and 'SQ_VALU_MFMA_BUSY_CYCLES' in d:
        cyc = d['GRBM_GUI_ACTIVE'] / 8
        d['mfma_busy_frac_of_simd_cycles'] = d['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024)
    if 'SQ_LDS_IDX_ACTIVE' in d and 'GRBM_GUI_ACTIVE' in d:
        d['lds_active_frac_of_cu_cycles'] = d['SQ_LDS_IDX_ACTIVE'] / (d['GRBM_GUI_ACTIVE'] / 8 * 256)
    if 'SQ_LDS_BANK_CONFLICT' in d and d.get('SQ_LDS_IDX_ACTIVE'):
        d['lds_conflict_frac'] = d['SQ_LDS_BANK_CONFLICT'] / d['SQ_LDS_IDX_ACTIVE']
    out[k] = {c: (round(v, 4) if v < 10 else int(v)) for c, v in d.items()}
# per kernel family (the names of bench.py's roofline): matrix-pipe busy fraction = sum of MFMA-busy cycles / (sum of dispatch cycles x 1024 SIMDs)
import sys
sys.path.insert(0, '$R/tools')
from pmc_traffic import family
fam = collections.defaultdict(lambda: [0.0, 0.0, 0])
for f in glob.glob('$R/gpurun_out/pmcs_${tag}_*/**/*counter_collection.csv', recursive=True):
    pass
raw = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob('$R/gpurun_out/pmcs_${tag}_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        fm = family(r['Kernel_Name'])
        if fm and r['Counter_Name'] in ('GRBM_GUI_ACTIVE', 'SQ_VALU_MFMA_BUSY_CYCLES'):
            a = raw[fm][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
fams = {}
for fm, cs in raw.items():
    if 'GRBM_GUI_ACTIVE' in cs and 'SQ_VALU_MFMA_BUSY_CYCLES' in cs and cs['GRBM_GUI_ACTIVE'][1] and cs['SQ_VALU_MFMA_BUSY_CYCLES'][1]:
        cyc = cs['GRBM_GUI_ACTIVE'][0] / cs['GRBM_GUI_ACTIVE'][1] / 8            # per dispatch, per XCD
        busy = cs['SQ_VALU_MFMA_BUSY_CYCLES'][0] / cs['SQ_VALU_MFMA_BUSY_CYCLES'][1]
        fams[fm] = {'mfma_busy': round(busy / (cyc * 1024), 4), 'dispatches_sampled': cs['GRBM_GUI_ACTIVE'][1]}
out['_families'] = fams
json.dump(out, open('$R/gpurun_out/pmcs_${tag}.json', 'w'), indent=1)
print('families', fams)
for k, d in sorted(((k_, d_) for k_, d_ in out.items() if not k_.startswith('_')), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0) * kv[1].get('dispatches', 0)):
    print(k, {c: d[c] for c in ('dispatches', 'GRBM_GUI_ACTIVE', 'mfma_busy_frac_of_simd_cycles', 'lds_active_frac_of_cu_cycles', 'lds_conflict_frac') if c in d})
PY
rm -rf $R/gpurun_out/pmcs_${tag}_[0-9]
