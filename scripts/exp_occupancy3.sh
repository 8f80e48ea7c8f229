# (a) pass B, two workgroups per CU against no bound, over the number of waves (fill of the last round)
# (b) pass A of 128-level slabs (one-wave workgroups): workgroups per CU  (c) pass A, 256 levels, wide short images
ab() { timeout -k 10 250 python -u scripts/ab_fused.py --steps 20 "$@" || exit 1; }
for w in 3328 3584 4352 5120 5632 6144; do echo "== (a) $w x 1080 x 256"; ab --width $w --height 1080 --levels 256 --opts '{}' '{"b_waves":8}' '{"a_lds_kb":54}' '{}'; done
for shape in "7680 4320" "3840 2160" "1920 1080"; do set -- $shape; echo "== (b) $1 x $2 x 128"; ab --width $1 --height $2 --levels 128 --opts '{}' '{"a_lds_kb":40}' '{"a_lds_kb":32}' '{"a_lds_kb":26}' '{"a_lds_kb":20}' '{"a_lds_kb":16}' '{}'; done
