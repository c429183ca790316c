# (historical: the persistent form of the four-wave groups, eval_values_loop_kernel + TWR_VALUES_LOOP, measured and not kept -- DESIGN 6.R5)
for l in 0 3 4 2 0 3; do echo "TWR_VALUES_LOOP=$l"; TWR_VALUES_LOOP=$l python scripts/values_c3.py 1 2>&1 | grep values_c3 | cut -c1-120; done
for l in 0 3 4; do echo "TWR_VALUES_LOOP=$l"; TWR_VALUES_LOOP=$l python scripts/planner_split.py 128 1024 2>&1 | grep "B=" | cut -c1-60; done
