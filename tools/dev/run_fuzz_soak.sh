# a long randomised soak (tests/test_gpu_fuzz.py) in a fresh region of the generators' seed space
export MZ_FUZZ_SEED_OFFSET=${1:-1}
export MZ_FUZZ_CASES=2500 MZ_FUZZ_CONV_CASES=1200 MZ_FUZZ_LEARN_CASES=1200 MZ_FUZZ_PLUMBING_CASES=600 MZ_FUZZ_SELFPLAY_CASES=400 MZ_FUZZ_EPILOGUE_CASES=500 MZ_FUZZ_GOMOKU_CASES=150 MZ_FUZZ_ATARI_CASES=60
timeout 3000 python -m pytest tests/test_gpu_fuzz.py -q -m gpu 2>&1 | tail -15
