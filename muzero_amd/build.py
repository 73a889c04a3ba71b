"""Builds libmzplanner_hip.so (planner kernels + C ABI) and libmzlearner_hip.so (learner-step kernels + C ABI) in-tree with hipcc
for gfx950.

    python -m muzero_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the numerical contract: all fused multiply-adds in
the kernels are explicit (see csrc/mz_device.h).
"""
import fcntl
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB_DIR = os.path.join(HERE, 'lib')
LIB_PATH = os.path.join(LIB_DIR, 'libmzplanner_hip.so')
SOURCES = ['planner.hip']
LEARNER_LIB_PATH = os.path.join(LIB_DIR, 'libmzlearner_hip.so')
LEARNER_SOURCES = ['learner.hip', 'learner_conv.hip', 'learner_replay.hip']
LEARNER_FILES = ('learner.hip', 'learner_conv.hip', 'learner_replay.hip', 'mz_learn.h', 'mz_learn_conv.h', 'mz_learn_conv_host.h', 'mzlearner.h')  # what only the learner library is compiled from (besides mz_device.h)
# -amdgpu-mfma-vgpr-form: MFMA accumulators in ordinary VGPRs where they fit.  The search kernels read every accumulator
# with VALU code right after the layer (ReLU, normalisation, partial sums); in the default AGPR form each of those reads is a
# v_accvgpr_read first (96 per simulation in k_search_fast: +2 % on C2, measured), and no kernel of this library needs the
# second register file (none spills either way).
# -amdgpu-sched-strategy=max-ilp: the search kernels run ONE wave per SIMD (LDS-bound occupancy), so the default strategy's
# occupancy-driven register economy buys nothing; scheduling for ILP does (+2 % C2, +1 % C3, conv kernels unchanged: measured).
FLAGS = ['--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-fPIC', '-shared', '-std=c++17', '-Wall', '-Wno-unused-function',
         '-Wno-pass-failed', '-mllvm', '-amdgpu-mfma-vgpr-form=1', '-mllvm', '-amdgpu-sched-strategy=max-ilp']


def _deps():
    """Every file the planner library is compiled from: csrc/ (sources and headers) without the learner's files, the public C
    header, this script."""
    return sorted(f for f in glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(CSRC, '*.hip')) if os.path.basename(f) not in LEARNER_FILES) + \
        [os.path.join(HERE, '..', 'include', 'mzplanner.h'), os.path.abspath(__file__)]


def _learner_deps():
    return [os.path.join(CSRC, 'learner.hip'), os.path.join(CSRC, 'learner_conv.hip'), os.path.join(CSRC, 'learner_replay.hip'), os.path.join(CSRC, 'mz_learn.h'), os.path.join(CSRC, 'mz_learn_conv.h'),
            os.path.join(CSRC, 'mz_learn_conv_host.h'), os.path.join(CSRC, 'mz_device.h'),
            os.path.join(HERE, '..', 'include', 'mzlearner.h'), os.path.abspath(__file__)]


def source_fingerprint():
    """sha256 over everything the kernels are compiled from (csrc/*, the C header, the compiler flags): what ties a committed
    profile (profiles/<round>/*/pmc_summary.json, written by tools/pmc_summary.py) to the build it was measured on.  bench.py
    reports `roofline.traffic` only from a summary whose fingerprint equals that of the sources it runs."""
    import hashlib

    h = hashlib.sha256()
    for d in _deps()[:-1]:  # (not this script: its flags are hashed below, its comments are not the build)
        h.update(os.path.basename(d).encode() + b'\0')
        with open(d, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()[:16]


def learner_fingerprint():
    """The same for libmzlearner_hip.so (profiles/<round>/learner/)."""
    import hashlib

    h = hashlib.sha256()
    for d in _learner_deps()[:-1]:
        h.update(os.path.basename(d).encode() + b'\0')
        with open(d, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in _deps())


def learner_needs_build():
    if not os.path.exists(LEARNER_LIB_PATH):
        return True
    t = os.path.getmtime(LEARNER_LIB_PATH)
    return any(os.path.getmtime(d) > t for d in _learner_deps())


def build_stamps(counters=False):
    """Diagnostic library with per-phase s_memtime stamps (never loaded by the product path); counters=True adds the
    atomic tree counters (hit rates, depths), which distort the timings."""
    hipcc = os.environ.get('HIPCC', 'hipcc')
    out = os.environ.get('MZ_STAMPS_OUT', os.path.join(LIB_DIR, 'libmzplanner_hip_stamps.so'))
    extra = ['-DMZ_STAMPS'] + (['-DMZ_COUNTERS'] if counters else []) + os.environ.get('MZ_EXTRA_FLAGS', '').split()
    subprocess.check_call([hipcc] + FLAGS + extra + [os.path.join(CSRC, s) for s in SOURCES] + ['-o', out])
    return out


def build(force=False, verbose=False):
    """Compile if any dependency is newer than the library.  Safe to call from several processes at once (ranks of a
    multi-GPU job, pytest-xdist workers): one holds the lock and compiles into a temporary file that is renamed into
    place, so nobody ever dlopen()s a partially written library."""
    if not force and not needs_build() and not learner_needs_build():
        return LIB_PATH
    hipcc = os.environ.get('HIPCC', 'hipcc')
    os.makedirs(LIB_DIR, exist_ok=True)
    with open(os.path.join(LIB_DIR, '.build.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            for need, path, sources, extra in ((needs_build, LIB_PATH, SOURCES, os.environ.get('MZ_EXTRA_FLAGS', '').split()),
                                               (learner_needs_build, LEARNER_LIB_PATH, LEARNER_SOURCES, [])):
                if not force and not need():  # up to date (or another process built it while we waited)
                    continue
                tmp = path + '.tmp.%d' % os.getpid()
                cmd = [hipcc] + FLAGS + extra + (['-Rpass-analysis=kernel-resource-usage'] if verbose else []) + \
                    [os.path.join(CSRC, s) for s in sources] + ['-o', tmp]
                try:
                    subprocess.check_call(cmd)
                    os.replace(tmp, path)
                finally:
                    if os.path.exists(tmp):
                        os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


def build_variant(out, extra):
    """A/B measurements (tools/ab_bench.py): the library with extra -D flags under another name; never loaded by the product path."""
    hipcc = os.environ.get('HIPCC', 'hipcc')
    subprocess.check_call([hipcc] + FLAGS + list(extra) + [os.path.join(CSRC, s) for s in SOURCES] + ['-o', out])
    return out


if __name__ == '__main__':
    if '--variant' in sys.argv:  # python -m muzero_amd.build --variant out.so -DX=1 ...
        i = sys.argv.index('--variant')
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:]))
        sys.exit(0)
    if '--stamps' in sys.argv:
        print(build_stamps(counters='--counters' in sys.argv))
        sys.exit(0)
    print(build(force='--force' in sys.argv, verbose='--verbose' in sys.argv))
