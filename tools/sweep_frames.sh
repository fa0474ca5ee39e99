cp swiftvideo_amd/libchipvideo.so /tmp/lib_orig.so
for v in base th32m6; do
  cp variants/$v.so swiftvideo_amd/libchipvideo.so
  for f in 1 4 16 64; do
    python bench.py --frames $f --steps 300 --warmup 20 --no-cpu-baseline --no-verify 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v frames=$f', round(d['ms_per_step']*1000,1), 'us', round(d['value'],1), 'Gpix/s')"
  done
done
cp /tmp/lib_orig.so swiftvideo_amd/libchipvideo.so
