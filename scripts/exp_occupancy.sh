# Resident workgroups per CU of passes A and B, bounded through an LDS floor (SMHIP_OPT_A_LDS_KB / _B_LDS_KB):
# 41 KiB -> 3 workgroups per CU, 54 -> 2, 81 -> 1.  A/B through scripts/ab_fused.py (events per pass, bit-equality).
ab() { timeout -k 10 250 python -u scripts/ab_fused.py --steps 20 "$@" || exit 1; }
echo "== 7680x4320x256"; ab --width 7680 --height 4320 --levels 256 --opts '{}' '{"b_lds_kb":54}' '{"b_lds_kb":54,"a_lds_kb":40}' '{"b_lds_kb":54,"a_lds_kb":54}' '{"b_lds_kb":54,"a_lds_kb":81}' '{}'
echo "== 3840x2160x512"; ab --width 3840 --height 2160 --levels 512 --opts '{}' '{"b_lds_kb":54}' '{"b_lds_kb":54,"a_lds_kb":54}' '{}'
echo "== 3072x1728x256"; ab --width 3072 --height 1728 --levels 256 --opts '{}' '{"b_lds_kb":54}' '{"b_lds_kb":54,"a_lds_kb":54}' '{"a_lds_kb":41}' '{}'
echo "== 2560x1440x256"; ab --width 2560 --height 1440 --levels 256 --opts '{}' '{"b_lds_kb":54}' '{"b_lds_kb":54,"a_lds_kb":54}' '{"a_lds_kb":41}' '{}'
echo "== 7680x4320x128"; ab --width 7680 --height 4320 --levels 128 --opts '{}' '{"paired":0,"b_lds_kb":54}' '{"paired":0,"b_lds_kb":54,"a_lds_kb":54}' '{"a_lds_kb":54}' '{"a_lds_kb":81}' '{}'
echo "== 3840x2160x128"; ab --width 3840 --height 2160 --levels 128 --opts '{}' '{"paired":0,"b_lds_kb":54}' '{"paired":0,"b_lds_kb":54,"a_lds_kb":54}' '{"a_lds_kb":54}' '{}'
echo "== 1920x1080x128"; ab --width 1920 --height 1080 --levels 128 --steps 50 --opts '{"paired":0}' '{"paired":0,"b_lds_kb":54}' '{"paired":0,"a_lds_kb":54}' '{"paired":0}'
echo "== 1920x1080x512"; ab --width 1920 --height 1080 --levels 512 --steps 50 --opts '{}' '{"b_lds_kb":54}' '{"b_lds_kb":54,"a_lds_kb":54}' '{}'
