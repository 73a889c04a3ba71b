#!/usr/bin/env python3
"""Copy the rocprofv3 summaries that tools/profile_all.sh wrote on the GPU box (gpurun_out/profiles/<workload>/) into the tracked
profiles/<round>/<workload>/ -- kernel_stats.csv (rocprofv3 --kernel-trace --stats) and pmc_summary.json (separate --pmc passes) --
and tie them to the build: the summary already carries the fingerprint of the kernel sources it was measured on
(muzero_amd.build.source_fingerprint, written by tools/pmc_summary.py on the box); here the git commit is added, and a summary whose
fingerprint differs from the working tree's sources is refused.  bench.py reports roofline.traffic only from a matching summary.

    python tools/stamp_profiles.py round3 c2 c3 c4 c5"""
import glob
import json
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from muzero_amd import build as mz_build  # noqa: E402


def main():
    rnd, workloads = sys.argv[1], sys.argv[2:]
    head = subprocess.check_output(['git', '-C', REPO, 'rev-parse', 'HEAD']).decode().strip()[:12]
    dirty = bool(subprocess.check_output(['git', '-C', REPO, 'status', '--porcelain', '--', 'muzero_amd/csrc', 'include']).decode().strip())
    fp = mz_build.source_fingerprint()
    for w in workloads:
        src = os.path.join(REPO, 'gpurun_out', 'profiles', w)
        doc = json.load(open(os.path.join(src, 'summary.json')))
        if w in ('learner', 'convlearner', 'atarilearner'):
            if doc.get('_learner_fingerprint') != mz_build.learner_fingerprint():
                sys.exit(f'learner: measured on other learner sources: re-run tools/profile_all.sh learner')
        elif doc.get('_source_fingerprint') != fp:
            sys.exit(f'{w}: measured on kernel sources {doc.get("_source_fingerprint")}, the working tree is {fp}: re-run tools/profile_all.sh')
        doc['_git_head'] = head + (' + uncommitted kernel changes' if dirty else '')
        dst = os.path.join(REPO, 'profiles', rnd, w)
        os.makedirs(dst, exist_ok=True)
        json.dump(doc, open(os.path.join(dst, 'pmc_summary.json'), 'w'), indent=1)
        stats = glob.glob(os.path.join(src, 'trace', '**', '*_kernel_stats.csv'), recursive=True)
        if stats:
            shutil.copy(stats[0], os.path.join(dst, 'kernel_stats.csv'))
        print(f'{w}: profiles/{rnd}/{w}/ <- kernel sources {fp}, commit {doc["_git_head"]}')


if __name__ == '__main__':
    main()
