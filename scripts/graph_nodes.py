"""List the nodes of a captured iteration graph in topological order from `bench.py --dump-graph PATH` (hipGraphDebugDotPrint
text): node kind, kernel name, how many of each -- so every launch of the iteration (and every memcpy / memset node torch's
capture added) is accounted for.    python scripts/graph_nodes.py gpurun_out/graph.dot [--all]"""
import collections, re, sys

txt = open(sys.argv[1]).read()
nodes, edges = {}, collections.defaultdict(list)
for m in re.finditer(r'"?(graph_\d+_node_\d+|node_\d+|\w+)"?\s*\[([^\]]*)\]', txt):
    name, attrs = m.group(1), m.group(2)
    lab = re.search(r'label\s*=\s*"([^"]*)"', attrs, re.S)
    if lab:
        nodes[name] = lab.group(1).replace("\\n", " | ").replace("\n", " | ")
for m in re.finditer(r'"?([\w]+)"?\s*->\s*"?([\w]+)"?', txt):
    edges[m.group(1)].append(m.group(2))
indeg = collections.Counter()
for a, bs in edges.items():
    for b in bs:
        indeg[b] += 1
order, q = [], [n for n in nodes if indeg[n] == 0]
while q:
    n = q.pop(0); order.append(n)
    for b in edges.get(n, []):
        indeg[b] -= 1
        if indeg[b] == 0 and b in nodes:
            q.append(b)


def short(label):
    k = re.search(r'(vln::\w+|__amd_rocclr_\w+|at::native::\w+|MEMCPY|MEMSET|EMPTY|memcpy|memset)', label)
    return k.group(1) if k else label[:60]


cnt = collections.Counter(short(nodes[n]) for n in order)
print(f"{len(order)} nodes")
for k, v in cnt.most_common():
    print(f"{v:4d}  {k}")
if "--all" in sys.argv:
    for i, n in enumerate(order):
        print(i, short(nodes[n]), "|", nodes[n][:160])
