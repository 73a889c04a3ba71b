export MZ_FUZZ_SEED_OFFSET=1
run() { echo "== $*"; env "$@" timeout 300 python tools/dev/atari_kf_probe.py 16 2>&1 | grep "^case\|conv_2.weight\|worst torch32" | cut -c1-150; }
run PROBE_K=6
run PROBE_K=1
run PROBE_K=5
run PROBE_SEED=1
run PROBE_SEED=2
run PROBE_CHAN=4
run PROBE_B=1
run PROBE_B=5
run PROBE_PLANES=16
run PROBE_RS=61
run PROBE_VS=601
