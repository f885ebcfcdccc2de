# sweep of the two knobs of a share's launch over parts on the 5 M job, 8 ranks (scripts/sim_scaling.py): candidates per round of
# the bar from which a row hands its remaining visits on (2^SG_HANDOVER_SHIFT) and the bar as a share of a wave's rounds
for cfg in ${SG_SWEEP:-"0 0.05" "1 0.07" "1 0.04" "1 0.05"}; do set -- $cfg; echo "== SG_HANDOVER_SHIFT=$1 SG_HEAVY_SHARE=$2"; SG_HANDOVER_SHIFT=$1 SG_HEAVY_SHARE=$2 python scripts/sim_scaling.py ${SG_SWEEP_ROWS:-5000000} f32 8 2>&1 | grep "^N=8" | cut -c1-420; done
