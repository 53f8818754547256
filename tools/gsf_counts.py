import sys, os
sys.path.insert(0, os.getcwd())
import bench, wittgenstein_amd as w
g = bench.make_sim(w, 4096, 0, 0, "gsf")
net = g.network()
tot = dict(delivered=0, tasks=0, events=0, draws=0)
rows = []
for t in range(400):
    net.runMs(1)
    s = net.last_stats
    rows.append((t, s["delivered"], s["tasks"], s["events"], s["draws"]))
for r in rows[100:140]:
    print("ms %d delivered %d tasks %d events %d draws %d" % r)
import numpy as np
a = np.array(rows)[100:400]
print("mean per ms over 100..400:", a[:,1:].mean(axis=0))
b = a[(a[:,0] % 10) != 1]
print("mean per ordinary ms:", b[:,1:].mean(axis=0))
