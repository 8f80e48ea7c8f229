# Pass B: resident waves per CU (SMHIP_OPT_B_WAVES) over shapes; A/B through scripts/ab_fused.py
ab() { timeout -k 10 250 python -u scripts/ab_fused.py --steps 20 "$@" || exit 1; }
L='{} {"b_waves":6} {"b_waves":8} {"b_waves":10} {"b_waves":12} {"b_waves":14} {}'
for shape in "1920 1080 256" "2560 1440 256" "3072 1728 256" "3840 2160 256" "1920 1080 512" "1280 720 256"; do
  set -- $shape; echo "== $1 x $2 x $3"; ab --width $1 --height $2 --levels $3 --opts $L
done
echo "== 7680 x 4320 x 256"; ab --width 7680 --height 4320 --levels 256 --opts '{}' '{"b_waves":8}' '{"b_waves":10}' '{"b_waves":12}' '{"b_waves":8,"keybuf":36}' '{"b_waves":10,"keybuf":28}'
echo "== 7680 x 4320 x 128"; ab --width 7680 --height 4320 --levels 128 --opts '{}' '{"b_waves":6}' '{"b_waves":8}' '{"b_waves":10,"keybuf":30}' '{"b_waves":4}'
echo "== 3840 x 2160 x 128"; ab --width 3840 --height 2160 --levels 128 --opts '{}' '{"b_waves":6}' '{"b_waves":8}' '{"b_waves":10,"keybuf":30}' '{"b_waves":4}'
