# pass A store cache policy (0 default, 2 nt, 16 sc1, 18 sc1+nt) in the real pipeline, several shapes
O="--opts {} {\"store_policy\":16} {\"store_policy\":2} {\"store_policy\":18}"
timeout -k 10 200 python scripts/ab_fused.py $O 2>&1 | tail -8 &&
echo "4K x 256" && timeout -k 10 200 python scripts/ab_fused.py --width 3840 --height 2160 --levels 256 --steps 10 $O 2>&1 | tail -4 &&
echo "8K x 256" && timeout -k 10 200 python scripts/ab_fused.py --width 7680 --height 4320 --levels 256 --steps 6 $O 2>&1 | tail -4 &&
echo "8K x 128" && timeout -k 10 200 python scripts/ab_fused.py --width 7680 --height 4320 --levels 128 --steps 6 $O 2>&1 | tail -4 &&
echo "8K x 128 of 1024 at 512" && timeout -k 10 200 python scripts/ab_fused.py --width 7680 --height 4320 --levels 128 --total 1024 --first 512 --steps 6 $O 2>&1 | tail -4
