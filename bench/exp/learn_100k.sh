# the reference's full training budget (num_steps_train = 100 000 updates) on the built-in Pendulum
set -e
O=gpurun_out/l100k; mkdir -p $O
for m in d4pg d3pg; do
  timeout -k 10 400 python bench/learn_pendulum.py --model $m --dtype bf16 --updates 100000 --out $O/$m > $O/$m.json 2> $O/$m.err
  tail -c 420 $O/$m.json; echo
done
