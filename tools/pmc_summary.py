#!/usr/bin/env python3
"""Summarise the rocprofv3 outputs written by tools/profile_all.sh for one workload directory:
kernel stats (top kernels by total time) + per-kernel means of every collected PMC counter, with the HBM byte
conversion MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE in KiB counting 64 B per 128-B request: doubled;
WRITE_SIZE in KiB taken as is)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d, targs, pargs = sys.argv[1], sys.argv[2], sys.argv[3]
    prog = sys.argv[4] if len(sys.argv) > 4 else 'bench.py'
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    from muzero_amd import build as mz_build

    head = None
    try:
        import subprocess

        head = subprocess.check_output(['git', '-C', repo, 'rev-parse', 'HEAD'], stderr=subprocess.DEVNULL).decode().strip()[:12]
    except Exception:  # the GPU box gets a snapshot without .git: tools/stamp_profiles.py adds the commit when the summary is copied into profiles/
        pass
    out = {'_source_fingerprint': mz_build.source_fingerprint(), '_learner_fingerprint': mz_build.learner_fingerprint(), '_git_head': head,
           '_trace_command': f'rocprofv3 --kernel-trace --stats --output-format csv -- python3 {prog} {targs}',
           '_pmc_command': f'rocprofv3 --kernel-trace --output-format csv --pmc <group> -- python3 {prog} {pargs}  (one run per counter group)'}
    stats = glob.glob(os.path.join(d, 'trace', '**', '*_kernel_stats.csv'), recursive=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        out['kernel_stats'] = [dict(name=r['Name'], calls=int(r['Calls']), total_ms=float(r['TotalDurationNs']) / 1e6, avg_us=float(r['AverageNs']) / 1e3,
                                    pct=float(r['Percentage'])) for r in rows[:12]]
    per = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, 'pmc*', '**', '*_counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name'].split('(')[0]
            per[name][r['Counter_Name']].append(float(r['Counter_Value']))
    pm = {}
    for k, cs in per.items():
        if not k.startswith(('void mz::', 'mz::', 'void mzl::', 'mzl::', 'void mzlc::', 'mzlc::')):
            continue
        e = {c: dict(launches=len(v), mean_per_launch=sum(v) / len(v)) for c, v in cs.items()}
        if 'FETCH_SIZE' in cs and 'WRITE_SIZE' in cs:
            f = 2.0 * 1024.0 * sum(cs['FETCH_SIZE']) / len(cs['FETCH_SIZE'])
            w = 1024.0 * sum(cs['WRITE_SIZE']) / len(cs['WRITE_SIZE'])
            e['_hbm_bytes_per_launch'] = dict(fetch_corrected_x2=f, write=w, total=f + w)
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in cs and 'GRBM_GUI_ACTIVE' in cs:
            # SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs:
            # util = busy / (1024 * gui / 8)
            e['_mfma_util'] = (sum(cs['SQ_VALU_MFMA_BUSY_CYCLES']) / len(cs['SQ_VALU_MFMA_BUSY_CYCLES'])) / (
                128.0 * sum(cs['GRBM_GUI_ACTIVE']) / len(cs['GRBM_GUI_ACTIVE']))
        if 'SQ_LDS_BANK_CONFLICT' in cs and 'SQ_LDS_IDX_ACTIVE' in cs and sum(cs['SQ_LDS_IDX_ACTIVE']) > 0:
            e['_lds_conflict_share'] = sum(cs['SQ_LDS_BANK_CONFLICT']) / sum(cs['SQ_LDS_IDX_ACTIVE'])
        pm[k] = e
    out['pmc'] = pm
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
