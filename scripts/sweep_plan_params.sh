for S in 12 16 20; do for C in 512 1024 2048; do for SH in 64 128 256; do
python bench.py --steps 10 --warmup 3 --slices $S --chunk $C --short $SH --no-cpu-baseline --no-backward 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('S=$S chunk=$C short=$SH ms', round(r['ms_per_step'],3))"
done; done; done
