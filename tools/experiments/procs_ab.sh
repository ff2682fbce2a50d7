# Run ON THE GPU BOX: the same GPU driven by ONE process or by TWO (torch.distributed.run, both ranks on device 0, gloo), chains per process swept
cd "$GRAFT_REPO_ROOT"
F="--steps 40 --warmup 4 --device 0 --dist-backend gloo --no-single-chain --no-step-micro --no-cpu-baseline --no-survey-size --no-step-circuit --no-batch128 --no-whole-pbs --no-ivc"
port=29600
for cfg in ${CFGS:-"2 3 64" "2 3 0" "2 6 0" "2 6 64" "3 4 64" "4 3 64"}; do
  set -- $cfg; port=$((port+1))
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus $1 --chains $2 --device-witness $3 $F 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); n=$1*$2; print('procs=$1 chains/proc=$2 dw=$3', 'value %.4f'%d['value'], 'ms/proof %.2f'%(d['ms_per_step']/n), 'load %.0f'%d['host']['loadavg_1min'])"
done
