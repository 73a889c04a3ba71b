# every example in every mode, briefly: crash / hang check
out=gpurun_out/examples; mkdir -p $out
run() { name=$1; shift; timeout 240 python "$@" > $out/$name.log 2>&1; rc=$?; echo "$name rc=$rc: $(grep -v amdgpu.ids $out/$name.log | tail -1 | cut -c1-220)"; }
run cartpole_hip examples/train_cartpole.py --train-steps 800 --report-every 400
run cartpole_hip_overlap examples/train_cartpole.py --train-steps 800 --report-every 400 --overlap
run cartpole_graphed examples/train_cartpole.py --train-steps 800 --report-every 400 --learner graphed
run cartpole_eager examples/train_cartpole.py --train-steps 400 --report-every 200 --learner eager
run cartpole_host examples/train_cartpole.py --train-steps 400 --report-every 200 --learner eager --host-assembly
run tictactoe examples/train_tictactoe.py --train-steps 800 --report-every 400 --eval-games 5
run tictactoe_host examples/train_tictactoe.py --train-steps 400 --report-every 200 --eval-games 5 --host-assembly
run gomoku examples/train_gomoku.py --train-steps 300 --report-every 150 --eval-games 2
run gomoku_graphed examples/train_gomoku.py --train-steps 300 --report-every 150 --eval-games 2 --graphed
run gomoku_host examples/train_gomoku.py --train-steps 300 --report-every 150 --eval-games 2 --host-assembly
run gomoku15 examples/train_gomoku.py --train-steps 150 --report-every 150 --eval-games 1 --board 15 --envs 64 --graphed
