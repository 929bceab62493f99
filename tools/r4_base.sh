#!/bin/bash
# round-4 baseline: stage times of every bench config (GPU box, repo root)
O=gpurun_out/${1:-r4a}; mkdir -p $O
for c in 1080p webcam 4k pacman; do
  python bench.py --steps 20 --profile-all --no-extras --config $c > $O/base_$c.json 2>> $O/base.err || exit 1
done
python bench.py --steps 5 --profile-all --no-extras --config 8k > $O/base_8k.json 2>> $O/base.err || exit 1
FUSE=1 python tools/stage_times.py > $O/fw_stage.txt 2>> $O/base.err
python tools/stage_times.py >> $O/fw_stage.txt 2>> $O/base.err
for c in 1080p webcam 4k pacman 8k; do python -c "
import json,sys; d=json.load(open('$O/base_$c.json')); print('$c', d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['roofline']['frac'])"; done
cat $O/fw_stage.txt
