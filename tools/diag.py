import ctypes, os, sys
sys.path.insert(0,'/root/repo')
import linne_amd
print("count before torch:", linne_amd.device_count())
c = linne_amd.lib.LINNEAmd_ContextCreate(0, 1<<28)
print("ctx", c)
import torch
print("torch cuda", torch.cuda.is_available(), torch.version.hip)
os.system("cat /proc/%d/maps | grep -i amdhip | awk '{print $6}' | sort -u" % os.getpid())
