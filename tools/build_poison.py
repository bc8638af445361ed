"""Debug builds of librovat_hip.so whose LDS scratch starts as garbage (-DRV_POISON_LDS=<pattern>; build/librovat_poison_*.so):
    python tools/build_poison.py   # here; then on the GPU box: bash tools/gpu.sh <tag> poison"""
import sys, threading
sys.path.insert(0,'/root/repo')
from robovat_amd import lib
def b(name, flag):
    lib.compile_lib('/root/repo/build/librovat_%s.so' % name, extra=[flag]); print('built', name, flush=True)
ts=[threading.Thread(target=b,args=a) for a in (('poison_nan','-DRV_POISON_LDS=0x7fc00000'),('poison_big','-DRV_POISON_LDS=0x7f7fffff'),('poison_rnd','-DRV_POISON_LDS=0x3f9d7a31'))]
# (an odd pattern is hashed per word, an even one is the same in every word.)  A second set: python tools/build_poison.py more
if len(sys.argv) > 1 and sys.argv[1] == 'more':
    ts=[threading.Thread(target=b,args=a) for a in (('poison_neg','-DRV_POISON_LDS=0xe846f640'),('poison_max','-DRV_POISON_LDS=0x7f7ffffe'),('poison_two','-DRV_POISON_LDS=0x00000002'))]
[t.start() for t in ts]; [t.join() for t in ts]
