# The library's occupancy policy ({}) against no bounds ({"b_waves":0,"a_lds_kb":0}) over shapes
ab() { timeout -k 10 250 python -u scripts/ab_fused.py --steps 20 "$@" || exit 1; }
O='{"b_waves":0,"a_lds_kb":0}'
for shape in "1920 1080 256" "1920 1080 512" "1920 1080 128" "1920 1080 64" "2560 1440 256" "3072 1728 256" "3840 2160 256" "3840 2160 512" "3840 2160 128" "5120 2880 256" "7680 4320 128" "7680 4320 256"; do
  set -- $shape; echo "== $1 x $2 x $3"; ab --width $1 --height $2 --levels $3 --opts '{}' $O '{}' $O
done
echo "== 1920 x 1080 x 512, a_lds"; ab --width 1920 --height 1080 --levels 512 --opts '{}' '{"a_lds_kb":54}' '{"a_lds_kb":41}' '{"a_lds_kb":27}'
echo "== 3840 x 2160 x 512, a_lds"; ab --width 3840 --height 2160 --levels 512 --opts '{}' '{"a_lds_kb":54}' '{"a_lds_kb":41}' '{"a_lds_kb":27}' '{"a_lds_kb":0}'
echo "== 7680 x 4320 x 1024"; ab --width 7680 --height 4320 --levels 1024 --steps 5 --opts '{}' '{"a_lds_kb":54}' '{"a_lds_kb":27}' '{"a_lds_kb":0}' $O
