#!/bin/bash
# usage: kexp2.sh <file to touch> <EXTRA flag> <pattern>
cd /root/repo/t1k_amd/csrc && touch $1 && make -j8 all EXTRA="$2" > /dev/null 2>&1
bash /root/repo/tools/scratch/kexp.sh "$3"
