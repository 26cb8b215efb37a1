for m in "exact:1" "split_bf16x3_train:1" "split_bf16x3_train:2" "split_bf16x3_train:3"; do
  mode=${m%%:*}; sw=${m##*:}
  VY_SPLIT_TRAIN=$sw python bench.py --mode train --conv-mode $mode --no-pmc --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$mode VY_SPLIT_TRAIN=$sw', round(r['value'],1), 'fps', round(r['ms_per_step'],2), 'ms')
"
done
