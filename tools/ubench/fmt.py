import re, sys
txt = open(sys.argv[1]).read().split('\n')
i = 0
while i < len(txt):
    l = txt[i]
    if l and not l.startswith(' '):
        print(l)
    elif 'us/launch' in l:
        m = re.match(r'\s+(.*?)\s+([\d.]+) us/launch.*err ([\d.e+-]+)', l)
        extra = ''
        if i + 1 < len(txt) and 'stamps' in txt[i+1]:
            s = txt[i+1]
            mm = re.search(r'issue ([\d.]+) first-land ([\d.]+) loop ([\d.]+) lds\+barrier ([\d.]+) finish ([\d.]+) \| realtime: entry skew ([\d.]+), first entry -> last exit ([\d.]+) \(first exit ([\d.]+)\), longest wave ([\d.]+)', s)
            if mm:
                extra = ' iss %s loop %s red %s fin %s | skew %s span %s long %s' % (mm.group(1), mm.group(3), mm.group(4), mm.group(5), mm.group(6), mm.group(7), mm.group(9))
        if m:
            print('  %-40s %7s %s%s' % (m.group(1), m.group(2), m.group(3), extra))
        else:
            print(l)
    i += 1
