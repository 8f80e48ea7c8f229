cd /root/repo
for o in '{}' '{"multi_txt":64}' '{"multi_txt":64,"iter_groups":2}' '{"multi_txt":64,"iter_groups":8}' '{"iter_groups":1}' '{"iter_groups":2}' '{"iter_groups":8}' '{"multi_txt":64,"tseg":32,"iter_groups":1}' '{"multi_txt":64,"tseg":32,"iter_groups":2}'; do
  echo "== $o"; timeout -k 10 100 python -u scripts/bench_exp2.py 1920 1080 256 "$o" 2>/dev/null | grep "smhip_min_pad_wta_keys  "
done
