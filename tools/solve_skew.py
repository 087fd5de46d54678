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
