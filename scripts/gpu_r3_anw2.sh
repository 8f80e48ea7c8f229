cd /root/repo
for shape in "1920 1080 512 9" "1280 720 256 9" "2048 1536 256 9" "1920 1080 256 5" "1920 1080 256 7" "1920 1080 256 11" "1920 1080 256 3" "2560 1440 256 9" "960 540 256 9" "1920 1080 1024 9"; do
  set -- $shape
  echo "== $1 x $2 x $3 window $4"
  timeout -k 10 100 python scripts/ab_fused.py --width $1 --height $2 --levels $3 --window $4 --steps 30 --opts '{}' '{"a_nw": 1}' 2>/dev/null | grep -v amdgpu | cut -c1-110
done
