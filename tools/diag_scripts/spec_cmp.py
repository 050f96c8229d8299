a=[l.split() for l in open("gpurun_out/spec_check_head.txt") if "digest" in l]
b=[l.split() for l in open("gpurun_out/spec_check.txt") if "digest" in l]
print("cases", len(a), len(b), "digests equal to the HEAD build:", sum(x[0]==y[0] and x[-1]==y[-1] for x,y in zip(a,b)))
for x,y in zip(a,b):
    if x[-1]!=y[-1]: print("  differs:", x[0])
