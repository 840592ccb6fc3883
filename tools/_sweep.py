import sys
sys.argv=[sys.argv[0],"none"]
exec(open("tools/pw2_probe.py").read())
for F in (32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512):
    gemm_case(F, 288, 288, 196, 1, cfgs=((0,-1,0),))
