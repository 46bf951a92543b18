#!/bin/bash
# round 4 end: 2048-high row blocks on the shard where >= 16 blocks per rank remain — one rank and two ranks sharing the card
O=gpurun_out/r04_nb2048_shard; mkdir -p $O
for nb in 1024 2048 1024 2048; do
  GPX_NB_SHARD=$nb timeout -k 10 500 python tools/shard_ab.py --ranks 2 --reps 2 > $O/ab_$nb.txt 2> $O/ab.err; echo "nb=$nb"; grep -a "^sharded\|^group" $O/ab_$nb.txt | cut -c1-170
done
