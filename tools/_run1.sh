for c in 0 64 96 128 192 256; do
P2P_FRONT_CHUNK=$c python bench.py --steps 10 --warmup 3 --no-legs 2>/dev/null | tail -1 | python -c "import json,sys,os; d=json.loads(sys.stdin.read()); print('front_chunk', os.environ.get('P2P_FRONT_CHUNK'), 'value', round(d['value']), 'ms', round(d['ms_per_step'],2))"
done
