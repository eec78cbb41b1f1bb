"""
How often does a CORRECT fp32 implementation land on the far side of C1's dominant kink?  (CPU only, test tooling.)

Config 1 (RealNVP moons-2D, K = 32, B = 256, weight seed 0): the flat gradient's distance to float64 is bimodal -- ~5e-3 .. 2e-2 when
every early-step ReLU decision agrees with float64, ~0.16 when ONE near-zero unit of an early step falls on the other side (the backward
pass amplifies that sample's O(1 / B) perturbation ~1.3 x per step).  This tool runs the oracle's fp32 train step on 48 row permutations
of the batch (a symmetry of the exact problem, a different rounding order) with five equally valid formulations of training-mode
BatchNorm -- torch's own F.batch_norm (statistics accumulated in double on the CPU), fp32 statistics, the fused x * sc + (b - mean * sc)
form and the subtract-first form -- and prints the fraction of runs beyond 0.1 / 0.03 / 0.012.

Measured in the authoring container (profiles/r05_c1_kink_odds.txt): F.batch_norm itself is on the far side in 58 % of the runs, the other
formulations in 19 .. 40 %; on the GPU box's EPYC 9575F the same F.batch_norm oracle was there in 1 run of 7 (profiles/r04_fullsize_parity.txt)
and the GPU path in 5 of 7.  The odds belong to the (implementation, host) pair, not to its correctness: tests/test_gpu_fullsize_parity.py
therefore compares DISTRIBUTIONS over row permutations and treats a bimodal yard-stick as bimodal.
"""
import importlib, sys, numpy as np, torch
from types import SimpleNamespace as NS
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nets as onets, trajectory as traj
import torch.nn.functional as F
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
nfdata = importlib.import_module('normalizing-flows-pytorch_amd.data')
torch.set_num_threads(1)
torch.manual_seed(0); np.random.seed(0)
net = pkg.RealNVP((2,), '2d', NS(layers=32, mixtures=None))
y = nfdata.sample('moons', 256, 1234)
sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
def flat(g, r64):
    num=den=0.0
    for k,e in r64['grads'].items():
        if k in g:
            d=g[k].double().reshape(-1)-e.double().reshape(-1); num+=float(d@d); den+=float(e.double().reshape(-1)@e.double().reshape(-1))
    return (num/den)**0.5
r64 = traj.run('realnvp',(2,),'2d',32,sd,y,1,dtype=torch.float64)[0][1]
gp = torch.Generator().manual_seed(99)
perms=[torch.arange(256)]+[torch.randperm(256,generator=gp) for _ in range(47)]
orig = onets.batch_norm
def run_all(tag):
    out=[]
    for pm in perms:
        r = traj.run('realnvp',(2,),'2d',32,sd,y[pm],1,dtype=torch.float32)[0][1]
        out.append(flat(r['grads'], r64))
    print(tag, '| n %d median %.2e  frac > 0.1: %.2f  frac > 0.03: %.2f  frac > 0.012: %.2f' % (len(out), np.median(out), np.mean(np.array(out)>0.1), np.mean(np.array(out)>0.03), np.mean(np.array(out)>0.012)), flush=True)
run_all('F.batch_norm (double-accumulated stats)      ')
def bn_fp32(x, sd_, p, training):
    w, b = sd_[p+'weight'], sd_[p+'bias']
    if x.dtype != torch.float32 or not training:
        return orig(x, sd_, p, training)
    mean = x.mean(0)
    var = ((x-mean)**2).mean(0)
    return (x-mean)/torch.sqrt(var+1e-5)*w+b
onets.batch_norm = bn_fp32
run_all('fp32 stats, (x-mean)/sqrt(var+eps)*w+b         ')
def bn_fp32_fused(x, sd_, p, training):
    w, b = sd_[p+'weight'], sd_[p+'bias']
    if x.dtype != torch.float32 or not training:
        return orig(x, sd_, p, training)
    mean = x.mean(0)
    var = ((x-mean)**2).mean(0)
    sc = w/torch.sqrt(var+1e-5)
    return x*sc + (b-mean*sc)
onets.batch_norm = bn_fp32_fused
run_all('fp32 stats, x*sc + (b - mean*sc)               ')
def bn_dbl_fused(x, sd_, p, training):
    w, b = sd_[p+'weight'], sd_[p+'bias']
    if x.dtype != torch.float32 or not training:
        return orig(x, sd_, p, training)
    xd=x.double()
    mean = xd.mean(0); var=((xd-mean)**2).mean(0)
    mean=mean.float(); invstd=(1/torch.sqrt(var+1e-5)).float()
    sc = w*invstd
    return x*sc + (b-mean*sc)
onets.batch_norm = bn_dbl_fused
run_all('double stats rounded, x*sc + (b - mean*sc)     ')
def bn_dbl_sub(x, sd_, p, training):
    w, b = sd_[p+'weight'], sd_[p+'bias']
    if x.dtype != torch.float32 or not training:
        return orig(x, sd_, p, training)
    xd=x.double()
    mean = xd.mean(0); var=((xd-mean)**2).mean(0)
    mean=mean.float(); invstd=(1/torch.sqrt(var+1e-5)).float()
    return (x-mean)*(w*invstd)+b
onets.batch_norm = bn_dbl_sub
run_all('double stats rounded, (x-mean)*(w*invstd)+b    ')
