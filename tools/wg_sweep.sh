for n in 1 2 3 4 5; do
  CMPC_WG_PER_CU=$n python3 bench.py --no-cpu-baseline --no-extras --batch 20480 --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('wg_per_cu $n', round(d['outcome']['all_instances_per_s']), round(d['roofline']['kernel_ms'],1))"
done
