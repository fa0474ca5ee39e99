#!/bin/bash
# tools/upload_sweep.sh — the upload-inclusive cfg2 leg (bench.py --with-upload) by frames per H2D copy (1 / 4 / 8 / 16) and upload streams (1 / 2).  GPU box.
for g in 1 4 8 16; do for st in 1 2; do
python bench.py --with-upload --upload-group $g --upload-streams $st --steps 5 --warmup 2 --min-seconds-other 0.6 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('group',$g,'streams',$st,'h2d',round(c['h2d_GBps_per_gpu'],1),'GB/s', 'Gpix/s', round(d['value'],2), 'node', c['pinned_numa_node'], 'verified', c['verified_vs_oracle'])"
done; done
python tools/h2d_probe.py 2>&1 | tail -8
