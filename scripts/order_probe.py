import sys
sys.path.insert(0, '.')
order = sys.argv[1]
if order == "lib_first":
    from stereomatch_amd import capi
    c = capi.Context(0)
    import torch
    print("lib first: torch sees", torch.cuda.is_available(), torch.cuda.device_count())
else:
    import torch
    print("torch first:", torch.cuda.is_available(), torch.cuda.device_count())
    from stereomatch_amd import capi
    c = capi.Context(0)
    print("ctx ok")
import ctypes
print([l.split()[-1] for l in open('/proc/self/maps') if 'amdhip64' in l and 'r-xp' in l])
