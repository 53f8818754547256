import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import bench, wittgenstein_amd as w
sims, batch = bench.make_batch(w, 32768, range(4), 0, 4, "handel")
batch.run_multiple_times(chunk=10, maxTime=20000)
for g in sims:
    sc = g.network().read("sigsChecked")
    print("sigsChecked max %d mean %.1f  time %d" % (sc.max(), sc.mean(), g.network().time))
