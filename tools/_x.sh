for k in 8 20 40 8 20 40; do
python bench.py --steps $k --warmup 1 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print($k, d['value'], d['ms_per_step'])"
done
