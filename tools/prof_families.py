import re, collections, sys
fam = collections.Counter(); cnt = collections.Counter()
for line in open(sys.argv[1]):
    if line.startswith("#"): continue
    m = re.search(r"calls\s+(\d+)\s+total_us\s+([\d.]+)", line)
    if not m: continue
    name = line[:64]
    key = re.sub(r"<.*", "", name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void zs::", "")).split("(")[0].strip()
    fam[key] += float(m.group(2)); cnt[key] += int(m.group(1))
for k, v in fam.most_common(8):
    print("%-44s %8.1f us/step  %6.1f launches/step  avg %.1f us" % (k, v / 11, cnt[k] / 11, v / cnt[k]))
print("total", sum(fam.values()) / 11)
