"""developer (ON THE GPU BOX): per-phase times of the populated-rows backward on the cfg4 rooms (36 -> 13, or ci co args),
instrumentation build -DCONV3P_SP_ABLATE=128.  usage: CONV3P_HIP_LIB=devlibs/lib_t_base.so python tools/phase_trace_room.py [ci co stride]"""
import os, sys, re, collections, subprocess
here = os.path.dirname(os.path.abspath(__file__))
args = sys.argv[1:] if len(sys.argv) > 1 else ["36", "13", "1"]
out = subprocess.run([sys.executable, os.path.join(here, "bw_trace.py"), args[0], args[1], "16", "4096", "room", args[2]],
                     capture_output=True, text=True).stdout
out = out.split("---- backward")[-1]
rows = [l for l in out.splitlines() if l.startswith("bsp<")]
acc = collections.OrderedDict()
for l in rows:
    for k, v in re.findall(r"([A-Za-z+\-]+) (\d+)(?= |$)", l.split(":", 1)[1]):
        acc.setdefault(k, []).append(int(v))
print("%d waves; mean / max (us; rounds: count): " % len(rows) + "  ".join("%s %.1f/%.1f" % (k, sum(v) / len(v) / (1 if k == "rounds" else 100), max(v) / (1 if k == "rounds" else 100)) for k, v in acc.items()))
