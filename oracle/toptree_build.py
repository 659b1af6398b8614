"""CPU restatement (plain Python, small inputs) of the reference's force-tree build under a domain decomposition.

TEST INFRASTRUCTURE ONLY.  Follows /root/reference/libgadget/forcetree.cpp:
  force_tree_create_topnodes / force_create_node_for_topnode   :651-690, 868-930   (the complete top-level grid, pseudo leaves)
  get_subnode, init_internal_node                             :277-328
  add_particle_to_tree, create_new_node_layer, modify_internal_node, add_particle_moment_to_node   :361-520, 945-966
  force_update_node_recursive (empty daughters removed unless TopLevel), force_update_particle_node :985-1101
  force_update_node_parallel (over this task's top leaves)     :1118-1141
  force_exchange_pseudodata, force_treeupdate_pseudos           :1136-1281
The top tree is given as geometry (daughter TopNode per octant), as the C-ABI takes it: the Peano-Hilbert numbering of the
daughters (`sub`, :881) only decides which TopNodes entry sits in which octant and stays with the caller.
No fixture of the reference covers this build (its tests run one task): parity unpinned, restated line by line; single-task
results equal the pinned single-domain builder (tests/test_toptree_build_cpu.py)."""
import numpy as np

NMAXCHILD = 8
NODEFULL = 1 << 16
PARTICLE, NODE, PSEUDO = 0, 1, 2


class N:
    __slots__ = ("center", "len", "sibling", "father", "suns", "nocc", "TopLevel", "InternalTopLevel", "ChildType", "cofm", "mass", "hmax", "DependsOnLocalMass")

    def __init__(self):
        self.center = [0.0, 0.0, 0.0]
        self.len = 0.0
        self.sibling = -10
        self.father = -10
        self.suns = [-1] * NMAXCHILD
        self.nocc = 0
        self.TopLevel = 0
        self.InternalTopLevel = 0
        self.ChildType = PARTICLE
        self.cofm = [0.0, 0.0, 0.0]
        self.mass = 0.0
        self.hmax = 0.0
        self.DependsOnLocalMass = 0


class Tree:
    def __init__(self, firstnode, lastnode):
        self.firstnode, self.lastnode = firstnode, lastnode
        self.nodes = {}
        self.nnext = firstnode

    def new(self):
        no = self.nnext
        self.nnext += 1
        self.nodes[no] = N()
        return no


def init_internal_node(t, child, parent, subnode):
    """forcetree.cpp:302-328"""
    c, p = t.nodes[child], t.nodes[parent]
    lenhalf = 0.25 * p.len
    c.len = 0.5 * p.len
    for j in range(3):
        sign = 1 if (subnode & (1 << j)) else -1
        c.center[j] = p.center[j] + sign * lenhalf


def get_subnode(node, pos):
    """forcetree.cpp:277-283"""
    return int(pos[0] > node.center[0]) + (int(pos[1] > node.center[1]) << 1) + (int(pos[2] > node.center[2]) << 2)


def create_topnodes(geo, leaf_task, ThisTask, Box, firstnode, lastnode):
    """force_tree_create_topnodes + force_create_node_for_topnode; geo[t] = (daughter[8], leaf).  Returns tree, leaf_treenode."""
    t = Tree(firstnode, lastnode)
    root = t.new()
    r = t.nodes[root]
    r.len = Box * 1.001
    r.center = [Box / 2.0] * 3
    r.father = -1
    r.sibling = -1
    r.TopLevel = 1
    leaf_treenode = [-1] * len(leaf_task)
    if geo[0][0][0] < 0:
        leaf_treenode[geo[0][1]] = root

    def rec(no, topnode):
        daughters = geo[topnode][0]
        if daughters[0] < 0:
            return
        kids = []
        for count in range(8):
            c = t.new()
            kids.append(c)
            p = t.nodes[no]
            p.suns[count] = c
            p.InternalTopLevel = 1
            p.ChildType = NODE
            p.nocc = NODEFULL
            init_internal_node(t, c, no, count)
            cn = t.nodes[c]
            cn.father = no
            cn.TopLevel = 1
            ct = daughters[count]
            if geo[ct][0][0] < 0:
                leaf = geo[ct][1]
                leaf_treenode[leaf] = c
                cn.suns[0] = leaf + lastnode
                if leaf_task[leaf] != ThisTask:
                    cn.ChildType = PSEUDO
        p = t.nodes[no]
        for j in range(7):
            t.nodes[p.suns[j]].sibling = p.suns[j + 1]
        t.nodes[p.suns[7]].sibling = p.sibling
        for count in range(8):
            rec(kids[count], daughters[count])

    rec(root, 0)
    return t, leaf_treenode


def add_moment(node, pos, mass, hsml):
    """add_particle_moment_to_node, forcetree.cpp:945-966 (hsml None: not a gas / BH particle)"""
    node.mass += mass
    for k in range(3):
        node.cofm[k] += mass * pos[k]
    if hsml is not None:
        for j in range(3):
            node.hmax = max(node.hmax, abs(pos[j] - node.center[j]) + hsml - node.len / 2.0)


def insert(t, i, cur, P):
    """add_particle_to_tree + create_new_node_layer, forcetree.cpp:361-520; P = (pos, mass, hsml-or-None per particle)"""
    pos, mass, hs = P
    while True:
        node = t.nodes[cur]
        if node.nocc < NODEFULL:
            break
        cur = node.suns[get_subnode(node, pos[i])]
    node = t.nodes[cur]
    nocc = node.nocc
    node.nocc += 1
    if nocc < NMAXCHILD:
        node.suns[nocc] = i
        add_moment(node, pos[i], mass[i], hs[i])
        return
    parent = cur
    first = cur
    while True:
        pn = t.nodes[parent]
        old = list(pn.suns)
        new = []
        for s in range(8):
            c = t.new()
            new.append(c)
            init_internal_node(t, c, parent, s)
            t.nodes[c].father = parent
        for k in range(NMAXCHILD):
            sub = get_subnode(pn, pos[old[k]])
            ch = t.nodes[new[sub]]
            ch.suns[ch.nocc] = old[k]
            add_moment(ch, pos[old[k]], mass[old[k]], hs[old[k]])
            ch.nocc += 1
        pn.suns = new
        for s in range(7):
            t.nodes[new[s]].sibling = new[s + 1]
        t.nodes[new[7]].sibling = pn.sibling
        pn.cofm = [0.0, 0.0, 0.0]
        pn.mass = 0.0
        pn.hmax = 0.0
        sub = get_subnode(pn, pos[i])
        child = new[sub]
        ch = t.nodes[child]
        if ch.nocc < NMAXCHILD:
            ch.suns[ch.nocc] = i
            add_moment(ch, pos[i], mass[i], hs[i])
            ch.nocc += 1
            break
        ch.ChildType = NODE
        ch.nocc = NODEFULL
        parent = child
    t.nodes[first].ChildType = NODE
    t.nodes[first].nocc = NODEFULL


def update_particle_node(node):
    """force_update_particle_node, forcetree.cpp:985-1003"""
    if node.mass > 0:
        for j in range(3):
            node.cofm[j] /= node.mass
    else:
        node.cofm = list(node.center)


def update_recursive(t, no, sib):
    """force_update_node_recursive, forcetree.cpp:1016-1101"""
    node = t.nodes[no]
    suns = node.suns
    jj = 0
    out = []
    for j in range(8):
        while jj < 8 and not t.nodes[suns[jj]].TopLevel and t.nodes[suns[jj]].ChildType == PARTICLE and t.nodes[suns[jj]].nocc == 0:
            jj += 1
        out.append(suns[jj] if jj < 8 else -1)
        jj += 1
    node.suns = suns = out
    for j in range(8):
        p = suns[j]
        if p < 0:
            continue
        nextsib = sib
        for k in range(j + 1, 8):
            if suns[k] >= 0:
                nextsib = suns[k]
                break
        t.nodes[p].sibling = nextsib
        if t.nodes[p].ChildType == PARTICLE:
            update_particle_node(t.nodes[p])
        if t.nodes[p].ChildType == NODE:
            update_recursive(t, p, nextsib)
    for j in range(8):
        p = suns[j]
        if p < 0:
            continue
        c = t.nodes[p]
        node.mass += c.mass
        node.cofm[0] += c.mass * c.cofm[0]
        node.cofm[1] += c.mass * c.cofm[1]
        node.cofm[2] += c.mass * c.cofm[2]
        if c.hmax > node.hmax:
            node.hmax = c.hmax
    if node.mass > 0:
        for j in range(3):
            node.cofm[j] /= node.mass


def treeupdate_pseudos(t, no):
    """force_treeupdate_pseudos, forcetree.cpp:1211-1281"""
    node = t.nodes[no]
    if not node.InternalTopLevel:
        return
    for j in range(8):
        p = node.suns[j]
        if t.nodes[p].InternalTopLevel:
            treeupdate_pseudos(t, p)
    node.mass = 0.0
    node.cofm = [0.0, 0.0, 0.0]
    node.hmax = 0.0
    for j in range(8):
        c = t.nodes[node.suns[j]]
        node.mass += c.mass
        node.cofm[0] += c.mass * c.cofm[0]
        node.cofm[1] += c.mass * c.cofm[1]
        node.cofm[2] += c.mass * c.cofm[2]
        if c.hmax > node.hmax:
            node.hmax = c.hmax
        if c.DependsOnLocalMass:
            node.DependsOnLocalMass = 1
    if node.mass:
        for j in range(3):
            node.cofm[j] /= node.mass
    else:
        node.cofm = list(node.center)


def build(pos, mass, hsml, order, geo, leaf_task, ThisTask, Box, firstnode, lastnode):
    """force_tree_create_nodes for the particles `order` (indices, insertion order) + the local half of force_tree_calc_moments.
    hsml[i] is None for particles without a smoothing length.  Returns tree, leaf_treenode, local leaf moments."""
    t, leaf_treenode = create_topnodes(geo, leaf_task, ThisTask, Box, firstnode, lastnode)
    P = (pos, mass, hsml)
    for i in order:
        # the particle's top leaf: walk the top tree (the reference looks Part[i].TopLeaf up; geometrically the same cell)
        cur = t.firstnode
        while t.nodes[cur].InternalTopLevel:
            cur = t.nodes[cur].suns[get_subnode(t.nodes[cur], pos[i])]
        if t.nodes[cur].ChildType == PSEUDO:
            raise ValueError("Bad topleaf: particle %d in a top leaf of another task" % i)
        insert(t, int(i), cur, P)
    moments = [None] * len(leaf_task)
    for leaf, task in enumerate(leaf_task):
        no = leaf_treenode[leaf]
        node = t.nodes[no]
        if task != ThisTask:
            moments[leaf] = ([0.0, 0.0, 0.0], 0.0, 0.0)
            continue
        node.DependsOnLocalMass = 1
        if node.ChildType == NODE:
            update_recursive(t, no, node.sibling)
        elif node.ChildType == PARTICLE:
            update_particle_node(node)
        moments[leaf] = (list(node.cofm), node.mass, node.hmax)
    return t, leaf_treenode, moments


def finish(t, leaf_treenode, leaf_task, ThisTask, all_moments):
    """force_exchange_pseudodata (the gathered table applied to the other tasks' leaves) + force_treeupdate_pseudos"""
    for leaf, task in enumerate(leaf_task):
        if task == ThisTask:
            continue
        node = t.nodes[leaf_treenode[leaf]]
        node.cofm = list(all_moments[leaf][0])
        node.mass = all_moments[leaf][1]
        node.hmax = all_moments[leaf][2]
    treeupdate_pseudos(t, t.firstnode)


def preorder(t):
    """nodes in the order the threaded tree is walked (first daughter, else sibling): what shq_tree_download numbers them by.
    Yields (node number, node)."""
    no = t.firstnode
    while no >= 0:
        node = t.nodes[no]
        yield no, node
        if node.ChildType == NODE:
            no = node.suns[0]
        else:
            no = node.sibling
