import sys, time, os
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),'tests'), os.path.join(os.getcwd(),'circuitgen')]
import torch
import numpy as np, oracle as orc
from vpbs_amd import synth
w = synth.trace(1, 135, 15)
for _ in range(2):
    t=time.time(); b = orc.Batch(w,3,4,True); print("wires commit", round(time.time()-t,3), flush=True)
