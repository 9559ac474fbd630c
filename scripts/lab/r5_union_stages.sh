#!/bin/bash
# lab (round 5): LDS stages of vbs_union_f32_kernel by workgroups per CU.  (1) n true clusters x 48 rows (scripts/lab/r5_union_split.py) with the stages the launch picks and with two forced;
# (2) the benchmark set's clustered family under its true grouping (r5_union_l2.py): 3 workgroups per CU x 2 stages (default) against 2 x 3 and 1 x 6 (plans of SPARTA_UNION_WPC workers per CU)
echo "== clusters x 48 rows: default stages | SPARTA_UNION_STAGES=2"
python scripts/lab/r5_union_split.py 2>&1 | grep clusters | sed 's/split=1.*//'
SPARTA_UNION_STAGES=2 python scripts/lab/r5_union_split.py 2>&1 | grep clusters | sed 's/split=1.*//'
echo "== 2000 clusters: workgroups per CU x stages"
for cfg in 3:2 2:3 2:2 1:6; do w=${cfg%%:*}; s=${cfg#*:}; echo "wpc $w stages $s: $(SPARTA_UNION_WPC=$w SPARTA_UNION_STAGES=$s COLS=60000 PROBES=0,8 python scripts/lab/r5_union_l2.py 2>&1 | grep columns | cut -c30-70 | tr '\n' ' ')"; done
