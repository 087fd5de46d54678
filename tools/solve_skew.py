import sys, numpy as np
a = np.loadtxt(sys.argv[1])
pub, seen, cdone, xcc, gdone = a[:,1], a[:,2], a[:,3], a[:,4], a[:,5]
t0 = pub.min()
print("publish(8) spread us: min 0 max %.2f  p50 %.2f p90 %.2f" % ((pub.max()-t0)/100, (np.median(pub)-t0)/100, (np.quantile(pub,0.9)-t0)/100))
print("gathers(9) done  us rel last publish: min %.2f med %.2f max %.2f" % ((gdone.min()-pub.max())/100, (np.median(gdone)-pub.max())/100, (gdone.max()-pub.max())/100))
print("sums seen       us rel last publish: min %.2f med %.2f max %.2f" % ((seen.min()-pub.max())/100, (np.median(seen)-pub.max())/100, (seen.max()-pub.max())/100))
print("own publish -> own gathers done: med %.2f ; own gathers done -> sums seen: med %.2f max %.2f" % (np.median(gdone-pub)/100, np.median(seen-gdone)/100, (seen-gdone).max()/100))
late = np.argsort(pub)[-8:]
print("latest publishers (wg, us, xcc):", [(int(i), round((pub[i]-t0)/100,2), int(xcc[i])) for i in late])
for x in range(8):
    m = xcc == x
    if m.any(): print("xcc", x, "n", int(m.sum()), "publish med %.2f max %.2f" % ((np.median(pub[m])-t0)/100, (pub[m].max()-t0)/100), "lb range", int(a[m,0].min()), int(a[m,0].max()))

if a.shape[1] >= 9:      # pipelined solve (k_cgp_solve): the cycle of every workgroup, publish(8) -> publish(9), split into its stages
    pub9, tags, pre = a[:,6], a[:,7], a[:,8]
    cyc = (pub9 - pub) / 100
    print("cycle publish(8)->publish(9) us: min %.2f med %.2f max %.2f" % (cyc.min(), np.median(cyc), cyc.max()))
    order = np.argsort(pub)
    stages = [("tag wait", tags - pub), ("gather", gdone - tags), ("sums", seen - gdone), ("update+publish", pub9 - seen)]
    print("stage medians us (early third / late third): " + "   ".join("%s %.2f / %.2f" % (n, np.median(st[order[:85]]) / 100, np.median(st[order[-85:]]) / 100) for n, st in stages)
          + "   prefetch valid %.0f%% / %.0f%%" % (100 * pre[order[:85]].mean(), 100 * pre[order[-85:]].mean()))
    for i in order[-6:]:
        print("  late wg %d xcc %d: late by %.2f  tag wait %.2f gather %.2f sums %.2f update+publish %.2f prefetch %d" % (int(a[i,0]), int(xcc[i]), (pub[i]-t0)/100, (tags[i]-pub[i])/100, (gdone[i]-tags[i])/100, (seen[i]-gdone[i])/100, (pub9[i]-seen[i])/100, int(pre[i])))
