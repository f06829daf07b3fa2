"""Developer tool: one config-4 object-BA batch (8 objects) on the GPU, prints GPU ms and ms/iter."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointslot_amd import synth
from pointslot_amd.optimizer import Optimizer
nobj = int(sys.argv[1]) if len(sys.argv) > 1 else 8
survey = len(sys.argv) > 2 and sys.argv[2] == "survey"      # SURVEY 8d's own perturbation (the bench's metric_ba) instead of the small z-axis one
graphs = [synth.object_ba_problem(0x51070004 + j) if survey else synth.object_ba_problem(0x51070004 + j, perturb=(0.05, 1.0, 0.02), perturb_axis="z") for j in range(nobj)]
opt = Optimizer()
opt.ObjectLocalBundleAdjustment(graphs[:1])
for _ in range(2):
    r = opt.ObjectLocalBundleAdjustment(graphs)
    ms = opt.last_kernel_ms()
    it = max(x["iterations"] for x in r); tr = max(x["trials"] for x in r)
    print("objects %d: %.2f ms, %d iterations, %d trials -> %.3f ms/iter, %.3f ms/trial" % (nobj, ms, it, tr, ms / it, ms / tr))
