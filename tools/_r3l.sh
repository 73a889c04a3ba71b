out=gpurun_out/r3l; mkdir -p $out
for sd in 2 3; do timeout 400 python examples/train_cartpole.py --train-steps 15000 --envs 128 --seed $sd --eval-episodes 5 --report-every 2500 --out $out/learning_cartpole_s$sd.json > $out/train_cartpole_s$sd.log 2>&1; tail -3 $out/train_cartpole_s$sd.log; done
