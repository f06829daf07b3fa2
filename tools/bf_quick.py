"""Developer tool: SearchByBruceMatching on 1 / 8 / 64 objects x 1000 x 1000 descriptors (the a10 bench leg's problems); prints kernel ms."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointslot_amd import synth
from pointslot_amd.matcher import ORBmatcher
m = ORBmatcher(0.9, True)
for k in (1, 8, 64):
    probs = [synth.bruteforce_problem(0x51070100 + i, 1000, 1000) for i in range(k)]
    m.SearchByBruceMatching(probs[:1])
    for _ in range(3):
        m.SearchByBruceMatching(probs)
    print(k, "objects: kernel %.3f ms" % m.last_kernel_ms())
