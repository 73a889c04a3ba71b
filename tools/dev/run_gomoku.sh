out=gpurun_out/gomoku; mkdir -p $out
python -m pytest tests/test_gpu_learner.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python examples/train_gomoku.py --train-steps 3000 --envs 128 --report-every 1000 --graphed --out $out/learning_gomoku9_device_graphed.json 2>&1 | grep -v amdgpu.ids | tail -6
timeout 300 python examples/train_gomoku.py --train-steps 200 --envs 64 --board 15 --report-every 200 --eval-games 1 --out $out/gomoku15_smoke.json 2>&1 | grep -v amdgpu.ids | tail -3
