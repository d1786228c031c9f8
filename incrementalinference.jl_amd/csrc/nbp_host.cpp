// nbp_host.cpp -- native host side (include/nbp_host.h): graph, nested-dissection ordering, Bayes tree,
// clique potentials, Gibbs id lists, and the compilation of a whole-tree solve into a libnbp program.
// Pure host code; the arithmetic all happens in the kernels behind nbp_program_*.
//
// The algorithms are the ones of this repo's Python mirror (bayestree.py, solver.TreeProgram), which
// cite the reference file:line they restate; tests/test_native_host.py checks both give identical
// orders, cliques, schedules and descriptor bytes, and tests/test_tree_known_answers.py pins the Python
// side against the reference's own known answers.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <array>
#include <deque>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/nbp_host.h"

extern "C" nbp_status nbp_internal_fail(nbp_status code, const char *msg);
static nbp_status hfail(nbp_status c, const char *m) { return nbp_internal_fail(c, m); }

namespace {

struct HVar { int manifold; bool initialized, ismargin; };
struct HFac { nbp_factor_spec s; bool is_prior; int dens = -1; /* index among the graph's density slots: pass-through densities and sampler tables */ };
// a measurement component that is an AliasingScalarSampler (enum nbp_dist NBP_DIST_TABLE): its table lives in a slot of the
// factor's own, planned and written like the density of a pass-through prior
static inline bool spec_has_table(const nbp_factor_spec &s) {
  if (s.factor_kind == NBP_F_PASSTHROUGH) return false;
  for (int c = 0; c < s.ncomp && c < NBP_MAXC; c++)
    if (s.comp[c][12] == (double)NBP_DIST_TABLE) return true;
  return false;
}

inline int mani_dim(int m) { return m == NBP_SE2 ? 3 : (m == NBP_CIRCULAR ? 1 : m); }
inline int mani_P(int m) { return m == NBP_SE2 ? 6 : mani_dim(m); }

// seeds.py
inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
inline uint64_t op_seed(uint64_t base, uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
  uint64_t h = splitmix64(base);
  h = splitmix64(h ^ a);
  h = splitmix64(h ^ b);
  h = splitmix64(h ^ c);
  h = splitmix64(h ^ d);
  return h;
}
enum { NBP_DOWN_MCITERS = 3 };
enum { PASS_INIT = 0, PASS_UP = 1, PASS_DOWN = 2, PASS_UNIT = 3, PRODUCT_ID = 0xFFFF };

struct Clique {
  int id = 0, parent = 0;
  std::vector<int> frontals, seps, children, potentials, dwnPotentials, inmsg;
  std::vector<int> directPriorMsg, directvar, itervar, msgskip, directFrtlMsg;
  std::vector<int> upsched, dnsched;
  std::vector<int> upiter;  // iteration (1-based) of the fmcmc! call each up-schedule entry belongs to
  // useMsgLikelihoods (jointmsg.py): the clique sub graph during the up solve and the joint message sent up
  struct SubFac { char tag; int a, b; std::vector<int> vars; int tname; bool is_prior; int kind; };
  std::vector<SubFac> jf;                  // 'f': a = factor id | 'd': a = child, b = index | 'p': a = child, b = variable
  std::vector<std::array<int, 4>> rel;     // differential factors sent up: (sym1, sym2, type name, factor kind)
  std::vector<int> jpriors;                // separators that get a MsgPrior in the message
  bool hasPriors = false;
  std::vector<int> Dslot;                  // slot of the KDE of each differential factor
  std::vector<int> all() const {
    std::vector<int> a = frontals;
    a.insert(a.end(), seps.begin(), seps.end());
    return a;
  }
};

struct Stage { int kind; std::vector<char> bytes; int n; };

}  // namespace

struct nbp_graph {
  nbp_solver_params sp;
  std::vector<HVar> vars;
  std::vector<HFac> facs;
  std::vector<std::vector<int>> vfacs;  // variable -> factor ids, insertion order
  std::vector<int> dens_facs;           // PartialPriorPassThrough factors: each owns a density slot in every plan
  int init_dens0 = 0;                   // first density slot of the last graph-initialisation plan
  // graph initialisation (initAll!): last plan
  std::vector<int> init_vars;
  std::vector<Stage> init_stages;
  int init_slots = 0;
};

struct nbp_tree {
  const nbp_graph *g = nullptr;
  std::vector<int> order;
  std::vector<Clique> cl;  // cl[k-1] = clique k
  std::vector<int> frontal_of;  // variable -> clique id
  std::vector<int> roots;
  // slot plan
  int snapshot = 0, n_slots = 0, dens0 = 0;  // dens0: first of the pass-through density slots (after main / snap)
  std::vector<int> main_slot, snap_slot;
  std::vector<std::map<int, int>> B;  // per clique: variable -> slot
  std::vector<int> scratch, maxf;     // per clique: first scratch slot, widest product (one scratch row per step of a round)
  // compile products
  std::vector<Stage> stages;
  nbp_tree_stats st{};
  bool joint = false;   // NBP_SOLVER_MSG_LIKELIHOODS
  bool stored = false;  // NBP_SOLVER_STORED_MEASUREMENTS
  std::vector<std::vector<int>> jdnsched;  // joint mode: down schedule per clique
  // the schedules as they run (solver.TreeProgram._plan_symbolic): up = variables with a density, not marginalized;
  // rounds = the steps of a schedule grouped into sets of commuting steps (solver.TreeProgram._rounds)
  std::vector<std::vector<int>> usched, uiter, dsched;
  std::vector<std::vector<std::vector<int>>> urounds, drounds;
  std::map<std::array<int, 5>, uint64_t> meas_seed;  // (clique, tag, a, b, variable of a message) -> seed of the last fresh draw
  // multi-rank compile (solver.TreeProgram(owner=, rank=)): owner[c] = rank of clique c (index c - 1); this rank compiles
  // only its own cliques; tree edges that cross a rank boundary become exchange segments between the stage segments
  std::vector<int> owner;
  int rank = 0;
  std::vector<std::map<int, int>> ghost;   // per clique: variable -> landing slot of its up message (children on other ranks)
  struct Segment { int kind, first, last; std::vector<std::array<int, 2>> sends, recvs; };  // kind 0 = run [first, last), 1 = exchange (peer, slot)
  std::vector<Segment> segments;
  int seg_start = 0;
  bool mine(int cid) const { return owner.empty() || owner[cid - 1] == rank; }
  int own(int cid) const { return owner.empty() ? 0 : owner[cid - 1]; }
  // the slot a parent's MsgPrior reads for variable v of child `chd`
  int msg_slot(int chd, int v) const { return mine(chd) ? B[chd - 1].at(v) : ghost[chd - 1].at(v); }
};

namespace {

template <class T> bool contains(const std::vector<T> &v, const T &x) { return std::find(v.begin(), v.end(), x) != v.end(); }

// ---- nested dissection (bayestree.nestedDissectionOrder) -------------------------------------------
struct ND {
  int n;
  std::vector<std::set<int>> adj;
  std::vector<char> alive;  // not dense
  std::vector<std::vector<int>> components(const std::vector<int> &nodes) {
    std::set<int> left(nodes.begin(), nodes.end());
    std::vector<std::vector<int>> comps;
    while (!left.empty()) {
      int s = *left.begin();  // smallest index == min by insertion index
      std::set<int> comp{s};
      std::vector<int> stack{s};
      while (!stack.empty()) {
        int u = stack.back();
        stack.pop_back();
        for (int w : adj[u])
          if (left.count(w) && !comp.count(w)) { comp.insert(w); stack.push_back(w); }
      }
      for (int v : comp) left.erase(v);
      comps.emplace_back(comp.begin(), comp.end());  // sorted by index
    }
    return comps;
  }
  std::vector<std::vector<int>> bfs_levels(const std::vector<int> &nodes, int start) {
    std::set<int> in(nodes.begin(), nodes.end()), seen{start};
    std::vector<int> frontier{start};
    std::vector<std::vector<int>> levels;
    while (!frontier.empty()) {
      levels.push_back(frontier);
      std::vector<int> nxt;
      for (int u : frontier)
        for (int w : adj[u])  // std::set iterates in index order == sorted(adj[u], key=index)
          if (in.count(w) && !seen.count(w)) { seen.insert(w); nxt.push_back(w); }
      frontier = nxt;
    }
    return levels;
  }
};

std::vector<int> nested_dissection(const nbp_graph *g) {
  const int n = (int)g->vars.size();
  ND nd;
  nd.n = n;
  nd.adj.assign(n, {});
  for (const HFac &f : g->facs)
    for (int a = 0; a < f.s.nvars; a++)
      for (int b = 0; b < f.s.nvars; b++)
        if (f.s.vars[a] != f.s.vars[b]) nd.adj[f.s.vars[a]].insert(f.s.vars[b]);
  // dense nodes aside
  std::vector<int> degs;
  for (int v = 0; v < n; v++) degs.push_back((int)nd.adj[v].size());
  std::vector<int> sd = degs;
  std::sort(sd.begin(), sd.end());
  const int thr = sd.empty() ? 0 : std::max(16, 10 * sd[sd.size() / 2]);
  std::vector<int> dense;
  for (int v = 0; v < n; v++)
    if (degs[v] > thr) dense.push_back(v);
  std::vector<int> nodes0;
  if (!dense.empty() && (int)dense.size() < n) {
    std::set<int> ds(dense.begin(), dense.end());
    for (int v = 0; v < n; v++) {
      if (ds.count(v)) { nd.adj[v].clear(); continue; }
      for (int d : dense) nd.adj[v].erase(d);
      nodes0.push_back(v);
    }
  } else {
    dense.clear();
    for (int v = 0; v < n; v++) nodes0.push_back(v);
  }
  std::vector<int> order;
  struct Item { bool emit; std::vector<int> nodes; };
  std::vector<Item> stack;
  {
    auto comps = nd.components(nodes0);
    // Python: work = comps[::-1]; stack = [("split", c) for c in work]; pop from the end -> first component first
    for (auto it = comps.rbegin(); it != comps.rend(); ++it) stack.push_back({false, *it});
  }
  while (!stack.empty()) {
    Item it = std::move(stack.back());
    stack.pop_back();
    if (it.emit || it.nodes.size() <= 2) { order.insert(order.end(), it.nodes.begin(), it.nodes.end()); continue; }
    auto lv = nd.bfs_levels(it.nodes, it.nodes[0]);
    lv = nd.bfs_levels(it.nodes, lv.back()[0]);
    if (lv.size() < 3) { order.insert(order.end(), it.nodes.begin(), it.nodes.end()); continue; }
    std::vector<long> sizes;
    long acc = 0;
    for (auto &l : lv) { acc += (long)l.size(); sizes.push_back(acc); }
    const long total = sizes.back();
    long bestd = -1, bestl = -1;
    int bestk = -1;
    for (int k = 1; k + 1 < (int)lv.size(); k++) {
      const long left = sizes[k - 1], right = total - sizes[k];
      const long d = std::labs(left - right), l = (long)lv[k].size();
      if (bestk < 0 || d < bestd || (d == bestd && l < bestl)) { bestd = d; bestl = l; bestk = k; }
    }
    const std::vector<int> &sep = lv[bestk];
    std::set<int> ss(sep.begin(), sep.end());
    std::vector<int> rest;
    for (int v : it.nodes)
      if (!ss.count(v)) rest.push_back(v);
    stack.push_back({true, sep});
    auto comps = nd.components(rest);
    for (auto c = comps.rbegin(); c != comps.rend(); ++c) stack.push_back({false, *c});
  }
  order.insert(order.end(), dense.begin(), dense.end());
  return order;
}

// ---- Bayes net + tree (bayestree.buildBayesNet / buildTree) ---------------------------------------
void build_tree(nbp_tree *t) {
  const nbp_graph *g = t->g;
  const int n = (int)g->vars.size(), nf = (int)g->facs.size();
  std::vector<std::vector<int>> fvars(nf), vfacs = g->vfacs;
  for (int f = 0; f < nf; f++) fvars[f].assign(g->facs[f].s.vars, g->facs[f].s.vars + g->facs[f].s.nvars);
  std::vector<char> eliminated;
  eliminated.assign(nf, 0);
  std::vector<std::vector<int>> sep(n);
  for (int v : t->order) {
    std::vector<int> Si;
    for (size_t q = 0; q < vfacs[v].size(); q++) {
      const int f = vfacs[v][q];
      if (eliminated[f]) continue;
      for (int s : fvars[f])
        if (s != v && !contains(Si, s)) Si.push_back(s);
      eliminated[f] = 1;
    }
    sep[v] = Si;
    if (!Si.empty()) {  // the marginal over Si joins the graph
      const int name = (int)fvars.size();
      fvars.push_back(Si);
      eliminated.push_back(0);
      for (int s : Si) vfacs[s].push_back(name);
    }
  }
  std::vector<int> pos(n, 0);
  for (int i = 0; i < (int)t->order.size(); i++) pos[t->order[i]] = i;
  t->frontal_of.assign(n, 0);
  for (auto it = t->order.rbegin(); it != t->order.rend(); ++it) {
    const int var = *it;
    const std::vector<int> &Sj = sep[var];
    if (Sj.empty()) {
      Clique c;
      c.id = (int)t->cl.size() + 1;
      c.frontals = {var};
      t->cl.push_back(c);
      t->frontal_of[var] = c.id;
      t->roots.push_back(c.id);
      continue;
    }
    int felbl = Sj[0];
    for (int s : Sj)
      if (pos[s] < pos[felbl]) felbl = s;  // identifyFirstEliminatedSeparator
    const int cp = t->frontal_of[felbl];
    Clique &P = t->cl[cp - 1];
    std::vector<int> a = P.all(), b = Sj;
    std::sort(a.begin(), a.end());
    std::sort(b.begin(), b.end());
    if (a == b) {
      P.frontals.push_back(var);  // appendClique!
      t->frontal_of[var] = cp;
    } else {
      Clique c;
      c.id = (int)t->cl.size() + 1;
      c.frontals = {var};
      c.seps = Sj;
      c.parent = cp;
      t->cl[cp - 1].children.push_back(c.id);
      t->cl.push_back(c);
      t->frontal_of[var] = c.id;
    }
  }
}

std::vector<int> postorder(const nbp_tree *t) {
  std::vector<int> out;
  // iterative post-order, children in list order (bayestree.BayesTree.postorder)
  for (int r : t->roots) {
    std::vector<std::pair<int, size_t>> st{{r, 0}};
    while (!st.empty()) {
      auto &top = st.back();
      const Clique &c = t->cl[top.first - 1];
      if (top.second < c.children.size()) {
        const int ch = c.children[top.second++];
        st.push_back({ch, 0});
      } else {
        out.push_back(top.first);
        st.pop_back();
      }
    }
  }
  return out;
}

std::vector<int> stable_sort_by(const std::vector<int> &xs, const std::vector<int> &key) {
  std::vector<int> idx(xs.size());
  for (size_t i = 0; i < xs.size(); i++) idx[i] = (int)i;
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return key[a] < key[b]; });
  std::vector<int> out;
  for (int i : idx) out.push_back(xs[i]);
  return out;
}

nbp_status clique_potentials_and_ids(nbp_tree *t) {
  const nbp_graph *g = t->g;
  std::vector<char> used(g->facs.size(), 0);
  for (int cid : postorder(t)) {
    Clique &c = t->cl[cid - 1];
    const std::vector<int> cols = c.all();
    std::set<int> allv(cols.begin(), cols.end());
    // setCliqPotentials!
    std::vector<int> frtfcts;
    for (int fr : c.frontals)
      for (int f : g->vfacs[fr])
        if (!contains(frtfcts, f)) frtfcts.push_back(f);
    for (int f : frtfcts) {
      if (used[f]) continue;
      bool inside = true;
      for (int k = 0; k < g->facs[f].s.nvars; k++) inside &= allv.count(g->facs[f].s.vars[k]) > 0;
      if (inside) c.potentials.push_back(f);
    }
    for (int f : c.potentials) used[f] = 1;
    c.dwnPotentials = frtfcts;
    // compCliqAssocMatrices!
    for (int ch : c.children)
      for (int s : t->cl[ch - 1].seps) c.inmsg.push_back(s);
    const int nc = (int)cols.size(), nA = (int)c.potentials.size(), nM = (int)c.inmsg.size(), nfr = (int)c.frontals.size();
    std::vector<std::vector<int>> A(nA, std::vector<int>(nc, 0)), M(nM, std::vector<int>(nc, 0));
    for (int j = 0; j < nc; j++) {
      for (int i = 0; i < nM; i++) M[i][j] = (c.inmsg[i] == cols[j]);
      for (int i = 0; i < nA; i++) {
        const nbp_factor_spec &fs = g->facs[c.potentials[i]].s;
        for (int k = 0; k < fs.nvars; k++)
          if (fs.vars[k] == cols[j]) A[i][j] = 1;
      }
    }
    // setCliqMCIDs!
    auto colsum = [&](const std::vector<std::vector<int>> &X, int j) { int s = 0; for (auto &r : X) s += r[j]; return s; };
    std::vector<int> sumc(nc, 0), sumA(nc, 0), sumM(nc, 0), sumsr(nc, 0);
    long tot = 0;
    for (int j = 0; j < nc; j++) { sumA[j] = colsum(A, j); sumM[j] = colsum(M, j); sumc[j] = sumA[j] + sumM[j]; tot += sumc[j]; }
    auto rows_single = [&](const std::vector<std::vector<int>> &X) {
      for (auto &r : X) {
        int s = 0;
        for (int x : r) s += x;
        if (s == 1)
          for (int j = 0; j < nc; j++) sumsr[j] += r[j];
      }
    };
    rows_single(A);
    rows_single(M);
    for (int j = 0; j < nc; j++)
      if (sumsr[j] - sumc[j] == 0) c.directPriorMsg.push_back(cols[j]);
    for (int j = 0; j < nc; j++)
      if (sumc[j] == 1 && sumA[j] == 1) c.directvar.push_back(cols[j]);
    if (tot == 0) return hfail(NBP_ERR_ARG, "mcmcIterationIDs -- unaccounted variables");
    std::vector<int> usset = c.directvar;
    for (int j = 0; j < nc; j++)
      if (sumc[j] > 1 && !contains(usset, cols[j])) usset.push_back(cols[j]);
    std::vector<int> alliter;
    for (int v : usset)
      if (!contains(c.directPriorMsg, v)) alliter.push_back(v);
    std::vector<int> upmsg;
    for (int j = 0; j < nc; j++)
      if (sumM[j] >= 1) upmsg.push_back(cols[j]);
    std::vector<int> sing, nons;
    for (int v : alliter) (contains(upmsg, v) ? sing : nons).push_back(v);
    std::map<int, int> colidx;
    for (int j = 0; j < nc; j++) colidx[cols[j]] = j;
    auto keys = [&](const std::vector<int> &xs) { std::vector<int> k; for (int v : xs) k.push_back(sumc[colidx[v]]); return k; };
    nons = stable_sort_by(nons, keys(nons));
    sing = stable_sort_by(sing, keys(sing));
    c.itervar = nons;
    c.itervar.insert(c.itervar.end(), sing.begin(), sing.end());
    for (int j = nfr; j < nc; j++)
      if (sumc[j] == 1 && sumM[j] == 1) c.msgskip.push_back(cols[j]);
    for (int j = 0; j < nfr; j++)
      if (sumc[j] == 1 && sumM[j] == 1) c.directFrtlMsg.push_back(cols[j]);
  }
  return NBP_OK;
}

void schedules(nbp_tree *t) {
  const nbp_graph *g = t->g;
  const int iters = g->sp.gibbs_iters;
  for (Clique &c : t->cl) {
    auto fmcmc = [&](const std::vector<int> &l, int it) {
      if (l.size() == 1) it = 1;
      for (int k = 0; k < it; k++) {
        c.upsched.insert(c.upsched.end(), l.begin(), l.end());
        c.upiter.insert(c.upiter.end(), l.size(), k + 1);
      }
    };
    fmcmc(c.directFrtlMsg, 1);
    if (!c.msgskip.empty()) fmcmc(c.msgskip, 1);
    if (!c.itervar.empty()) fmcmc(c.itervar, iters);
    if (!c.directPriorMsg.empty()) {
      std::vector<int> l;
      for (int v : c.directPriorMsg)
        if (!contains(c.msgskip, v)) l.push_back(v);
      fmcmc(l, 1);
    }
    // down: determineCliqVariableDownSequence + solveCliqDownFrontalProducts!
    std::set<int> frs(c.frontals.begin(), c.frontals.end());
    std::vector<int> iterv;
    for (int f : c.dwnPotentials) {
      std::vector<int> hit;
      for (int k = 0; k < g->facs[f].s.nvars; k++)
        if (frs.count(g->facs[f].s.vars[k])) hit.push_back(g->facs[f].s.vars[k]);
      if (hit.size() > 1)
        for (int v : hit)
          if (!contains(iterv, v)) iterv.push_back(v);
    }
    std::vector<int> itf, directs;
    for (int v : c.frontals) (contains(iterv, v) ? itf : directs).push_back(v);
    if (g->sp.limitfixeddown) {
      auto drop = [&](std::vector<int> &l) { l.erase(std::remove_if(l.begin(), l.end(), [&](int v) { return g->vars[v].ismargin; }), l.end()); };
      drop(itf);
      drop(directs);
    }
    c.dnsched = directs;
    // MCIters = 3 is solveCliqDownFrontalProducts!'s own keyword default (CliqStateMachineUtils.jl:485); its only
    // caller (CliqueStateMachine.jl:838) does not pass gibbsIters
    for (int k = 0; k < NBP_DOWN_MCITERS; k++) c.dnsched.insert(c.dnsched.end(), itf.begin(), itf.end());
    if (c.parent == 0) c.dnsched.clear();
  }
}

// ---- compile (solver.TreeProgram) -----------------------------------------------------------------
// one density of a variable update: a factor ('f', a = factor id), the message prior of a child ('m', a = child
// clique) or a differential factor of a child's joint message ('d', a = child clique, b = index)
struct Entry {
  char tag; int a, b;
  Entry(bool msg, int ref) : tag(msg ? 'm' : 'f'), a(ref), b(0) {}
  Entry(char t, int a_, int b_) : tag(t), a(a_), b(b_) {}
};

int factor_type_name(const nbp_factor_spec &s) {  // type(fnc).__name__ of the Python mirror, as a code
  if (s.ncomp > 1) return 1000;                    // Mixture, whatever its mechanics
  return s.factor_kind * 2 + (s.partial_mask ? 1 : 0);
}
// selectFactorType (services/DefaultNodeTypes.jl:12-31): the default relative factor between two variables
bool select_factor_type(int m1, int m2, int *tname, int *kind) {
  if (m1 != m2) return false;
  if (m1 >= NBP_EUCLID1 && m1 <= NBP_EUCLID3) *kind = NBP_F_LINREL;
  else if (m1 == NBP_CIRCULAR) *kind = NBP_F_CIRCULAR;
  else if (m1 == NBP_SE2) *kind = NBP_F_SE2;
  else return false;
  *tname = *kind * 2;
  return true;
}

// findShortestPathDijkstra on the clique sub graph (unit weights, breadth first in the order of jointmsg.py):
// the factors along one shortest path; false when `to` cannot be reached.  tname >= 0: only such factors.
bool shortest_path_factors(const std::vector<int> &vars, const std::vector<Clique::SubFac> &facs, int frm, int to, int tname,
                           std::vector<int> *path) {
  path->clear();
  if (frm == to) return true;
  std::map<int, std::vector<int>> by_var;
  for (int v : vars) by_var[v];
  for (size_t i = 0; i < facs.size(); i++) {
    if (tname >= 0 && facs[i].tname != tname) continue;
    for (int v : facs[i].vars)
      if (by_var.count(v)) by_var[v].push_back((int)i);
  }
  std::map<int, std::pair<int, int>> prev;  // variable -> (previous variable, factor)
  prev[frm] = {-1, -1};
  std::deque<int> dq{frm};
  while (!dq.empty()) {
    const int v = dq.front();
    dq.pop_front();
    for (int i : by_var[v])
      for (int u : facs[i].vars)
        if (by_var.count(u) && !prev.count(u)) {
          prev[u] = {v, i};
          if (u == to) {
            for (int w = u; prev[w].first >= 0; w = prev[w].first) path->push_back(prev[w].second);
            std::reverse(path->begin(), path->end());
            return true;
          }
          dq.push_back(u);
        }
  }
  return false;
}

// plan_joint_messages (jointmsg.py): children before parents
void plan_joint(nbp_tree *t) {
  const nbp_graph *g = t->g;
  std::vector<int> height(t->cl.size() + 1, 0);
  for (int cid : postorder(t)) {
    int h = 0;
    const Clique &c = t->cl[cid - 1];
    if (!c.children.empty()) {
      for (int chd : c.children) h = std::max(h, height[chd]);
      h += 1;
    }
    height[cid] = h;
  }
  std::vector<int> ids;
  for (const Clique &c : t->cl) ids.push_back(c.id);
  std::stable_sort(ids.begin(), ids.end(), [&](int a, int b) { return height[a] != height[b] ? height[a] < height[b] : a < b; });
  for (int cid : ids) {
    Clique &c = t->cl[cid - 1];
    c.jf.clear(); c.rel.clear(); c.jpriors.clear();
    for (int f : c.potentials) {
      const nbp_factor_spec &fs = g->facs[f].s;
      c.jf.push_back({'f', f, 0, std::vector<int>(fs.vars, fs.vars + fs.nvars), factor_type_name(fs), g->facs[f].is_prior, fs.factor_kind});
    }
    for (int ch : c.children) {  // addMsgFactors!(subfg, msg, UpwardPass)
      const Clique &M = t->cl[ch - 1];
      for (size_t i = 0; i < M.rel.size(); i++)
        c.jf.push_back({'d', ch, (int)i, {M.rel[i][0], M.rel[i][1]}, M.rel[i][2], false, M.rel[i][3]});
      for (int v : M.jpriors) {  // addLikelihoodPriorCommon!
        bool touched = false;
        for (const auto &f : c.jf) touched |= contains(f.vars, v);
        if (M.hasPriors || !touched) c.jf.push_back({'p', ch, v, {v}, -2, true, NBP_F_MSGPRIOR});
      }
    }
    c.hasPriors = false;
    bool own_priors = false;
    for (const auto &f : c.jf) { c.hasPriors |= f.is_prior; own_priors |= (f.is_prior && f.tag == 'f'); }
    if (c.parent == 0) continue;
    const std::vector<int> allv = c.all();
    std::vector<int> dec = c.seps;  // sortperm(dims; rev = true), stable
    std::stable_sort(dec.begin(), dec.end(), [&](int a, int b) { return mani_dim(g->vars[a].manifold) > mani_dim(g->vars[b].manifold); });
    std::vector<int> acc(dec.rbegin(), dec.rend()), already, path;
    for (int s1 : dec) {
      already.push_back(s1);
      for (int s2 : acc) {
        if (contains(already, s2)) continue;
        if (!shortest_path_factors(allv, c.jf, s1, s2, -1, &path)) continue;
        std::vector<int> types;
        for (int fi : path)
          if (!contains(types, c.jf[fi].tname)) types.push_back(c.jf[fi].tname);
        int tn, kind;
        if (types.size() != 1 || !select_factor_type(g->vars[s1].manifold, g->vars[s2].manifold, &tn, &kind) || tn != types[0]) continue;
        c.rel.push_back({s1, s2, tn, kind});
      }
    }
    // _findSubgraphsFactorType + _generateSubgraphMsgPriors
    std::map<int, int> count, cls;
    for (int s_ : c.seps) count[s_] = 0;
    for (const auto &r : c.rel) { count[r[0]]++; count[r[1]]++; }
    int nw = 0;
    for (int s_ : c.seps)
      if (count[s_] == 0) cls[s_] = ++nw;
    std::vector<int> outer;
    for (int s_ : c.seps)
      if (!cls.count(s_)) outer.push_back(s_);
    for (int k1 : outer) {
      if (!cls.count(k1)) cls[k1] = ++nw;
      std::vector<int> inner;
      for (int s_ : c.seps)
        if (!cls.count(s_)) inner.push_back(s_);
      for (int k2 : inner) {
        int tn, kind;
        const bool sel = select_factor_type(g->vars[k1].manifold, g->vars[k2].manifold, &tn, &kind);
        const bool conn = sel && shortest_path_factors(allv, c.jf, k1, k2, tn, &path) && !path.empty();
        cls[k2] = conn ? cls[k1] : ++nw;
      }
    }
    std::map<int, std::vector<int>> classes;
    for (int s_ : c.seps) classes[cls[s_]].push_back(s_);
    for (const auto &kv : classes) {
      const std::vector<int> &syms = kv.second;
      if (!(syms.size() == 1 || own_priors)) continue;
      int md = 0;  // _calcCandidatePriorBest: highest dimension, then most factors in the sub graph
      for (int s_ : syms) md = std::max(md, mani_dim(g->vars[s_].manifold));
      int best = -1, bestadj = -1;
      for (int s_ : syms) {
        if (mani_dim(g->vars[s_].manifold) != md) continue;
        int adj = 0;
        for (const auto &f : c.jf) adj += contains(f.vars, s_) ? 1 : 0;
        if (adj > bestadj) { bestadj = adj; best = s_; }
      }
      c.jpriors.push_back(best);
    }
  }
}

// the densities of variable v in clique c: up solve (with the common message priors) or down solve (without)
std::vector<Entry> joint_entries(const Clique &c, int v, bool down) {
  // potentials and differentials in sub-graph order, the common message priors last (the reference iterates a Dict; with
  // this order a clique call -- factors, then messages -- numbers the densities of a variable the same way)
  std::vector<Entry> e;
  for (const auto &f : c.jf) {
    if (!contains(f.vars, v) || f.tag == 'p') continue;
    if (f.tag == 'f') e.push_back(Entry('f', f.a, 0));
    else e.push_back(Entry('d', f.a, f.b));
  }
  if (!down)
    for (const auto &f : c.jf)
      if (contains(f.vars, v) && f.tag == 'p') e.push_back(Entry('m', f.a, 0));
  return e;
}

void fill_proposal(const nbp_graph *g, nbp_proposal_desc &d, const HFac *fac, int msg_slot, int target, const std::map<int, int> *Bc,
                   const std::vector<int> *main_slot, const std::set<int> *inclq, int out_slot, uint64_t seed, double nullSurplus,
                   const std::vector<char> *isinit = nullptr, int keep_count = 0) {
  memset(&d, 0, sizeof(d));
  const nbp_solver_params &sp = g->sp;
  auto slot_of = [&](int v) { return Bc == nullptr ? v : ((inclq == nullptr || inclq->count(v)) ? Bc->at(v) : (*main_slot)[v]); };
  d.manifold = g->vars[target].manifold;
  d.out_slot = out_slot;
  d.inflate_cycles = sp.inflate_cycles;
  d.mhidx_in = d.mhidx_out = -1;
  d.spread_nh = sp.spread_nh;
  d.seed = seed;
  if (!fac) {  // MsgPrior (generateMsgPrior, TreeMessageUtils.jl:86-89)
    d.factor_kind = NBP_F_MSGPRIOR;
    d.inflation = sp.inflation;
    d.nullhypo = std::max(0.0, nullSurplus);
    d.nvars = 1;
    d.sfidx = 0;
    d.var_slot[0] = slot_of(target);
    d.var_slot[1] = msg_slot;
    d.ncomp = 1;
    d.comp[0][0] = 1.0;
    return;
  }
  const nbp_factor_spec &s = fac->s;
  if (s.factor_kind == NBP_F_PASSTHROUGH) {  // the density is the proposal (ApproxConv.jl:196-227); msg_slot = its slot
    d.factor_kind = NBP_F_PASSTHROUGH;
    d.partial_mask = s.partial_mask;
    d.inflation = s.inflation > 0 ? s.inflation : sp.inflation;
    d.nvars = 1;
    d.sfidx = 0;
    d.var_slot[0] = slot_of(target);
    d.var_slot[1] = msg_slot;
    d.ncomp = 1;
    d.comp[0][0] = 1.0;
    d.skip_bandwidth = 1;
    d.keep_count = keep_count;  // the only factor of its update: 1 = the belief keeps the density's point count, 2 = graph init
    return;
  }
  d.factor_kind = s.factor_kind;
  d.partial_mask = s.partial_mask;
  d.inflation = s.inflation > 0 ? s.inflation : sp.inflation;
  d.nullhypo = std::max(s.nullhypo, nullSurplus);
  d.nvars = s.nvars;
  for (int i = 0; i < s.nvars; i++) {
    if (s.vars[i] == target) d.sfidx = i;
    d.var_slot[i] = slot_of(s.vars[i]);
  }
  if (spec_has_table(s)) d.var_slot[NBP_MAXV - 1] = msg_slot;  // the slot of the sampler's table (callers pass it like a density slot)
  d.ncomp = s.ncomp;
  memcpy(d.comp, s.comp, sizeof(d.comp));
  if (s.has_multihypo) {
    int flags = 1 | 0x80;
    for (int i = 0; i < s.nvars; i++)
      if (isinit ? (*isinit)[s.vars[i]] != 0 : g->vars[s.vars[i]].initialized) flags |= 1 << (8 + i);
    d.has_multihypo = flags;
    for (int i = 0; i < s.nvars; i++) d.multihypo[i] = s.multihypo[i];
  }
}

void add_stage(nbp_tree *t, int kind, const void *descs, size_t esz, int n) {
  Stage st;
  st.kind = kind;
  st.n = n;
  st.bytes.assign((const char *)descs, (const char *)descs + esz * (size_t)n);
  t->stages.push_back(std::move(st));
}

// solver.TreeProgram._exchange: `edges` = (src rank, src slot or -1, dst rank, dst slot or -1) in a globally agreed order;
// a rank only knows the slots on its own side.  An empty copy stage makes libnbp flush the deferred bandwidth fits before
// slots travel; the stage list is cut there and an exchange segment recorded.
struct Edge { int src_rank, src_slot, dst_rank, dst_slot; };
void rank_exchange(nbp_tree *t, const std::vector<Edge> &edges) {
  nbp_tree::Segment x;
  x.kind = 1;
  x.first = x.last = 0;
  for (const Edge &e : edges) {
    if (e.src_rank == t->rank && e.dst_rank != t->rank) x.sends.push_back({e.dst_rank, e.src_slot});
    if (e.dst_rank == t->rank && e.src_rank != t->rank) x.recvs.push_back({e.src_rank, e.dst_slot});
  }
  if (x.sends.empty() && x.recvs.empty()) return;
  add_stage(t, NBP_STAGE_COPIES, nullptr, sizeof(nbp_copy_desc), 0);
  t->segments.push_back({0, t->seg_start, (int)t->stages.size(), {}, {}});
  t->segments.push_back(x);
  t->seg_start = (int)t->stages.size();
}
// up messages that cross a rank boundary: the child's separator beliefs (and, in joint-message mode, the KDEs of its
// differential factors) -> the parent rank's landing slots
void up_edges(const nbp_tree *t, const std::vector<int> &cliques, std::vector<Edge> &edges) {
  for (int cid : cliques) {
    const Clique &c = t->cl[cid - 1];
    const int so = t->own(cid), dn = t->own(c.parent);
    for (int v : c.seps)
      edges.push_back({so, so == t->rank ? t->B[cid - 1].at(v) : -1, dn, dn == t->rank ? t->ghost[cid - 1].at(v) : -1});
    if (t->joint)
      for (size_t i = 0; i < c.rel.size(); i++) {
        const int slot = (so == t->rank || dn == t->rank) ? c.Dslot[i] : -1;  // the sender's own slot / the landing slot
        edges.push_back({so, slot, dn, slot});
      }
  }
}

// the densities of variable v in clique c, up solve: clique potentials touching v + child messages on v
std::vector<Entry> up_entries(const nbp_tree *t, const Clique &c, int v) {
  if (t->joint) return joint_entries(c, v, false);
  const nbp_graph *g = t->g;
  std::vector<Entry> ent;
  for (int f : c.potentials) {
    bool hit = false;
    for (int q = 0; q < g->facs[f].s.nvars; q++) hit |= g->facs[f].s.vars[q] == v;
    if (hit) ent.push_back({false, f});
  }
  for (int chd : c.children)
    if (contains(t->cl[chd - 1].seps, v)) ent.push_back({true, chd});
  return ent;
}
// ... down solve: every factor of the frontal (addDownVariableFactors!)
std::vector<Entry> down_entries(const nbp_tree *t, const Clique &c, int v) {
  if (t->joint) return joint_entries(c, v, true);
  std::vector<Entry> ent;
  for (int f : t->g->vfacs[v]) ent.push_back({false, f});
  return ent;
}

// solver.TreeProgram._rounds: steps whose variables differ and share no factor commute (disjoint beliefs, random
// streams keyed by the step index); round[j] = 1 + the latest round among the earlier steps j conflicts with
template <typename EntriesOf>
std::vector<std::vector<int>> schedule_rounds(const nbp_tree *t, const std::vector<int> &sched, EntriesOf entries_of) {
  std::map<int, std::set<int>> reads;
  for (int v : sched) {
    if (reads.count(v)) continue;
    std::set<int> &r = reads[v];
    for (const Entry &e : entries_of(v)) {
      if (e.tag == 'f')
        for (int q = 0; q < t->g->facs[e.a].s.nvars; q++) r.insert(t->g->facs[e.a].s.vars[q]);
      else if (e.tag == 'd') {
        r.insert(t->cl[e.a - 1].rel[e.b][0]);
        r.insert(t->cl[e.a - 1].rel[e.b][1]);
      }
    }
    r.erase(v);
  }
  std::vector<int> rnd(sched.size(), 0);
  int top = -1;
  for (size_t j = 0; j < sched.size(); j++) {
    const int v = sched[j];
    for (size_t i = 0; i < j; i++) {
      const int u = sched[i];
      if (u == v || reads[v].count(u) || reads[u].count(v)) rnd[j] = std::max(rnd[j], rnd[i] + 1);
    }
    top = std::max(top, rnd[j]);
  }
  std::vector<std::vector<int>> groups(top + 1);
  for (size_t j = 0; j < sched.size(); j++) groups[rnd[j]].push_back((int)j);
  return groups;
}

void plan_rounds(nbp_tree *t) {
  const nbp_graph *g = t->g;
  const size_t nc = t->cl.size();
  t->usched.assign(nc, {});
  t->uiter.assign(nc, {});
  t->dsched.assign(nc, {});
  t->urounds.assign(nc, {});
  t->drounds.assign(nc, {});
  for (const Clique &c : t->cl) {
    const size_t k0 = c.id - 1;
    for (size_t k = 0; k < c.upsched.size(); k++) {  // doFMCIteration passes over marginalized variables and variables without a density
      const int v = c.upsched[k];
      if (!up_entries(t, c, v).empty() && !g->vars[v].ismargin) { t->usched[k0].push_back(v); t->uiter[k0].push_back(c.upiter[k]); }
    }
    t->dsched[k0] = t->joint ? t->jdnsched[k0] : c.dnsched;
    t->urounds[k0] = schedule_rounds(t, t->usched[k0], [&](int v) { return up_entries(t, c, v); });
    t->drounds[k0] = schedule_rounds(t, t->dsched[k0], [&](int v) { return down_entries(t, c, v); });
  }
}

nbp_status update_ops(nbp_tree *t, int cid, int v, const std::vector<Entry> &entries, const std::set<int> *inclq, int out_slot, int passid,
                      int step, uint64_t seed, std::vector<nbp_proposal_desc> &props, std::vector<nbp_product_desc> &prods,
                      bool fresh = true, int lane = 0) {
  const nbp_graph *g = t->g;
  const int base = t->scratch[cid - 1] + lane * t->maxf[cid - 1];  // `lane`: position of this step within its round
  const std::map<int, int> &Bc = t->B[cid - 1];
  const int F = (int)entries.size();
  if (F > NBP_MAXF) return hfail(NBP_ERR_RANGE, "a product exceeds NBP_MAXF densities");
  bool anymh = false;
  for (const Entry &e : entries)
    if (e.tag == 'f' && g->facs[e.a].s.has_multihypo) anymh = true;
  bool anypartial = false;
  nbp_product_desc q;
  memset(&q, 0, sizeof(q));
  int F_in = 0;
  for (int i = 0; i < F; i++) {
    const Entry &e = entries[i];
    HFac diff;  // 'd': LinearRelative(::MKD) & co. between the two separators, measurement = the child's KDE slot
    const HFac *fac = e.tag == 'f' ? &g->facs[e.a] : nullptr;
    if (e.tag == 'd') {
      const auto &r = t->cl[e.a - 1].rel[e.b];
      memset(&diff, 0, sizeof(diff));
      diff.is_prior = false;
      diff.s.factor_kind = r[3];
      diff.s.nvars = 2;
      diff.s.vars[0] = r[0];
      diff.s.vars[1] = r[1];
      diff.s.ncomp = 1;
      diff.s.comp[0][0] = 1.0;
      diff.s.comp[0][4] = diff.s.comp[0][8] = diff.s.comp[0][12] = 1.0;
      fac = &diff;
    }
    double ns = 0.0;  // _null_surplus: relative non-multihypo siblings of a multihypo factor
    if (anymh && fac && !fac->is_prior && !fac->s.has_multihypo) ns = g->sp.null_surplus_add;
    int msg_slot = e.tag == 'm' ? t->msg_slot(e.a, v) : -1;
    if (fac && fac->dens >= 0 && e.tag != 'm') msg_slot = t->dens0 + fac->dens;  // the slot of its density / sampler table
    nbp_proposal_desc d;
    const uint64_t sd = op_seed(seed, passid, cid, step, i + 1);
    fill_proposal(g, d, fac, msg_slot, v, &Bc, &t->main_slot, inclq, base + i, sd, ns, nullptr, F == 1 ? 1 : 0);
    if (e.tag == 'd') d.meas_kde = t->cl[e.a - 1].Dslot[e.b] + 1;
    {  // needFreshMeasurements (SolveTree.jl:119): one stored measurement per factor object
      const std::array<int, 5> key{cid, (int)e.tag, e.a, e.b, e.tag == 'm' ? v : -1};
      if (fresh) t->meas_seed[key] = sd;
      else {
        auto it = t->meas_seed.find(key);
        d.meas_seed = it == t->meas_seed.end() ? 0 : it->second;
      }
    }
    props.push_back(d);
    q.in_slot[i] = base + i;
    const int pm = fac ? fac->s.partial_mask : 0;
    q.in_partial[i] = (uint8_t)pm;
    anypartial |= pm != 0;
    F_in += (fac && fac->is_prior) ? 0 : 1;
  }
  const int man = g->vars[v].manifold;
  q.manifold = man;
  q.nfactors = F;
  q.niter = g->sp.product_niter;
  q.out_slot = out_slot;
  q.labels_out = -1;
  q.old_slot = -1;
  if (anypartial) q.old_slot = (inclq == nullptr || inclq->count(v)) ? Bc.at(v) : t->main_slot[v];
  else memset(q.in_partial, 0, sizeof(q.in_partial));
  q.seed = op_seed(seed, passid, cid, step, PRODUCT_ID);
  prods.push_back(q);
  const long N = g->sp.N, P = mani_P(man), D = mani_dim(man);
  t->st.alg_bytes += (F_in + 2) * N * P * 8 + (F_in + 1) * D * 8;
  t->st.alg_bytes_proposal += (F_in + 1) * N * P * 8;
  t->st.alg_bytes_prep += (F_in + 1) * D * 8;
  t->st.alg_bytes_product += N * P * 8;
  return NBP_OK;
}

}  // namespace

extern "C" {

nbp_status nbp_graph_create(const nbp_solver_params *p, nbp_graph **out) {
  if (!p || !out) return hfail(NBP_ERR_ARG, "null argument");
  if (p->N < 8 || p->N > NBP_MAXN) return hfail(NBP_ERR_RANGE, "N must be in [8, 512]");
  if (!p->upsolve && !p->downsolve) return hfail(NBP_ERR_ARG, "must attempt either up or down solve");
  nbp_graph *g = new nbp_graph();
  g->sp = *p;
  *out = g;
  return NBP_OK;
}
nbp_status nbp_graph_destroy(nbp_graph *g) {
  delete g;
  return NBP_OK;
}
int32_t nbp_graph_add_variable(nbp_graph *g, int32_t manifold) {
  if (!g) return hfail(NBP_ERR_ARG, "null argument");
  if (manifold < NBP_EUCLID1 || manifold > NBP_SE2) return hfail(NBP_ERR_ARG, "unknown manifold");
  g->vars.push_back({manifold, false, false});  // like addVariable!: not initialised until a belief is set
  g->vfacs.emplace_back();
  return (int32_t)g->vars.size() - 1;
}
int32_t nbp_graph_add_factor(nbp_graph *g, const nbp_factor_spec *s) {
  if (!g || !s) return hfail(NBP_ERR_ARG, "null argument");
  if (s->factor_kind < NBP_F_PRIOR || s->factor_kind > NBP_F_PASSTHROUGH || s->factor_kind == NBP_F_MSGPRIOR)
    return hfail(NBP_ERR_ARG, "factor: unknown kind");
  if (s->nvars < 1 || s->nvars > NBP_MAXV) return hfail(NBP_ERR_RANGE, "factor: nvars");
  if (s->ncomp < 1 || s->ncomp > NBP_MAXC) return hfail(NBP_ERR_RANGE, "factor: ncomp");
  const bool prior = s->factor_kind == NBP_F_PRIOR || s->factor_kind == NBP_F_PASSTHROUGH;
  if (prior && s->nvars != 1) return hfail(NBP_ERR_ARG, "priors are unary factors");
  if (s->factor_kind == NBP_F_PASSTHROUGH && s->has_multihypo) return hfail(NBP_ERR_ARG, "a pass-through prior takes no multihypo");
  for (int i = 0; i < s->nvars; i++)
    if (s->vars[i] < 0 || s->vars[i] >= (int)g->vars.size()) return hfail(NBP_ERR_RANGE, "factor: variable id");
  HFac f;
  f.s = *s;
  f.is_prior = prior;
  const int id = (int)g->facs.size();
  if (spec_has_table(*s) && s->nvars >= NBP_MAXV) return hfail(NBP_ERR_RANGE, "a factor with a sampler table takes at most NBP_MAXV - 1 variables");
  if (s->factor_kind == NBP_F_PASSTHROUGH || spec_has_table(*s)) { f.dens = (int)g->dens_facs.size(); g->dens_facs.push_back(id); }
  g->facs.push_back(f);
  for (int i = 0; i < s->nvars; i++) g->vfacs[s->vars[i]].push_back(id);
  return id;
}
nbp_status nbp_graph_set_variable_flags(nbp_graph *g, int32_t v, int32_t initialized, int32_t ismargin) {
  if (!g || v < 0 || v >= (int)g->vars.size()) return hfail(NBP_ERR_RANGE, "variable id");
  g->vars[v].initialized = initialized != 0;
  g->vars[v].ismargin = ismargin != 0;
  return NBP_OK;
}
int32_t nbp_graph_num_variables(const nbp_graph *g) { return g ? (int32_t)g->vars.size() : 0; }
int32_t nbp_graph_num_factors(const nbp_graph *g) { return g ? (int32_t)g->facs.size() : 0; }

nbp_status nbp_graph_order_nested_dissection(const nbp_graph *g, int32_t *out) {
  if (!g || !out) return hfail(NBP_ERR_ARG, "null argument");
  std::vector<int> o = nested_dissection(g);
  for (size_t i = 0; i < o.size(); i++) out[i] = o[i];
  return NBP_OK;
}

nbp_status nbp_tree_build(const nbp_graph *g, const int32_t *order, int32_t n, nbp_tree **out) {
  if (!g || !order || !out) return hfail(NBP_ERR_ARG, "null argument");
  if (n != (int)g->vars.size()) return hfail(NBP_ERR_ARG, "the elimination order must list every variable once");
  std::vector<char> seen(n, 0);
  for (int i = 0; i < n; i++) {
    if (order[i] < 0 || order[i] >= n || seen[order[i]]) return hfail(NBP_ERR_ARG, "the elimination order must list every variable once");
    seen[order[i]] = 1;
  }
  nbp_tree *t = new nbp_tree();
  t->g = g;
  t->order.assign(order, order + n);
  build_tree(t);
  nbp_status rc = clique_potentials_and_ids(t);
  if (rc) { delete t; return rc; }
  schedules(t);
  t->joint = (g->sp.flags & NBP_SOLVER_MSG_LIKELIHOODS) != 0;
  t->stored = (g->sp.flags & NBP_SOLVER_STORED_MEASUREMENTS) != 0;
  if (t->joint) {
    plan_joint(t);
    // no addDownVariableFactors! (CliqueStateMachine.jl:823): the down solve works on the clique sub graph as the
    // up solve left it, minus the common priors
    t->jdnsched.assign(t->cl.size(), {});
    for (Clique &c : t->cl) {
      if (c.parent == 0) continue;
      std::set<int> frs(c.frontals.begin(), c.frontals.end()), itv;
      for (const auto &f : c.jf) {
        if (f.tag == 'p') continue;
        int nfr = 0;
        for (int u : f.vars) nfr += frs.count(u) ? 1 : 0;
        if (nfr > 1)
          for (int u : f.vars)
            if (frs.count(u)) itv.insert(u);
      }
      auto skip = [&](int v) { return g->sp.limitfixeddown && g->vars[v].ismargin; };
      std::vector<int> &d = t->jdnsched[c.id - 1];
      for (int v : c.frontals)
        if (!itv.count(v) && !skip(v) && !joint_entries(c, v, true).empty()) d.push_back(v);
      for (int k = 0; k < NBP_DOWN_MCITERS; k++)
        for (int v : c.frontals)
          if (itv.count(v) && !skip(v)) d.push_back(v);
    }
  }
  plan_rounds(t);
  *out = t;
  return NBP_OK;
}
nbp_status nbp_tree_destroy(nbp_tree *t) {
  delete t;
  return NBP_OK;
}
int32_t nbp_tree_num_cliques(const nbp_tree *t) { return t ? (int32_t)t->cl.size() : 0; }
int32_t nbp_tree_max_schedule(const nbp_tree *t) {
  size_t m = 0;
  if (t)
    for (const Clique &c : t->cl) m = std::max(m, std::max(c.upsched.size(), c.dnsched.size()));
  return (int32_t)m;
}
nbp_status nbp_tree_clique(const nbp_tree *t, int32_t k, nbp_clique_info *info, int32_t *fr, int32_t *sp, int32_t *ch, int32_t *pots,
                           int32_t *up, int32_t *dn) {
  if (!t || k < 1 || k > (int)t->cl.size()) return hfail(NBP_ERR_RANGE, "clique id");
  const Clique &c = t->cl[k - 1];
  if (info) {
    info->parent = c.parent;
    info->nfrontals = (int)c.frontals.size();
    info->nseparators = (int)c.seps.size();
    info->nchildren = (int)c.children.size();
    info->npotentials = (int)c.potentials.size();
    info->nup = (int)c.upsched.size();
    info->ndown = (int)c.dnsched.size();
  }
  auto cp = [](const std::vector<int> &v, int32_t *o) { if (o) for (size_t i = 0; i < v.size(); i++) o[i] = v[i]; };
  cp(c.frontals, fr); cp(c.seps, sp); cp(c.children, ch); cp(c.potentials, pots); cp(c.upsched, up); cp(c.dnsched, dn);
  return NBP_OK;
}

nbp_status nbp_tree_clique_idlists(const nbp_tree *t, int32_t k, int32_t *counts, int32_t *dfm, int32_t *ms, int32_t *iv, int32_t *dpm) {
  if (!t || k < 1 || k > (int)t->cl.size()) return hfail(NBP_ERR_RANGE, "clique id");
  if (!counts) return hfail(NBP_ERR_ARG, "null argument");
  const Clique &c = t->cl[k - 1];
  auto cp = [](const std::vector<int> &v, int32_t *o) { if (o) for (size_t i = 0; i < v.size(); i++) o[i] = v[i]; };
  counts[0] = (int32_t)c.directFrtlMsg.size(); counts[1] = (int32_t)c.msgskip.size();
  counts[2] = (int32_t)c.itervar.size(); counts[3] = (int32_t)c.directPriorMsg.size();
  cp(c.directFrtlMsg, dfm); cp(c.msgskip, ms); cp(c.itervar, iv); cp(c.directPriorMsg, dpm);
  return NBP_OK;
}

int32_t nbp_tree_plan_slots(nbp_tree *t, int32_t snapshot) {
  if (!t) return hfail(NBP_ERR_ARG, "null argument");
  const nbp_graph *g = t->g;
  const int n = (int)g->vars.size();
  t->snapshot = snapshot;
  t->main_slot.resize(n);
  for (int v = 0; v < n; v++) t->main_slot[v] = v;
  int nxt = n;
  t->snap_slot.clear();
  if (snapshot) {
    t->snap_slot.resize(n);
    for (int v = 0; v < n; v++) t->snap_slot[v] = nxt + v;
    nxt += n;
  }
  t->dens0 = nxt;  // one slot per pass-through density, written by the caller like the initial beliefs
  nxt += (int)g->dens_facs.size();
  t->B.assign(t->cl.size(), {});
  t->ghost.assign(t->cl.size(), {});
  t->scratch.assign(t->cl.size(), 0);
  t->maxf.assign(t->cl.size(), 1);
  auto conc = [&](const Clique &c) {  // steps of one round run side by side: one scratch row each
    size_t m = 1;
    for (const auto &r : t->urounds[c.id - 1]) m = std::max(m, r.size());
    for (const auto &r : t->drounds[c.id - 1]) m = std::max(m, r.size());
    return (int)m;
  };
  for (Clique &c : t->cl) {  // clique ids ascending == Python's iteration over tree.cliques (insertion order)
    if (!t->mine(c.id)) continue;
    for (int v : c.all()) t->B[c.id - 1][v] = nxt++;
    for (int chd : c.children) {  // messages from children that live on another rank land in ghost slots
      if (t->mine(chd)) continue;
      Clique &cc = t->cl[chd - 1];
      for (int v : cc.seps) t->ghost[chd - 1][v] = nxt++;
      if (t->joint) {  // ... and so do the KDEs of its differential factors
        cc.Dslot.clear();
        for (size_t i = 0; i < cc.rel.size(); i++) cc.Dslot.push_back(nxt++);
      }
    }
    // widest product of this clique: up = potentials touching v + child messages on v; down = all factors of v
    size_t maxf = 1;
    if (t->joint) {
      c.Dslot.clear();
      for (size_t i = 0; i < c.rel.size(); i++) c.Dslot.push_back(nxt++);
      for (int v : c.upsched)
        if (!g->vars[v].ismargin) maxf = std::max(maxf, joint_entries(c, v, false).size());
      for (int v : t->jdnsched[c.id - 1]) maxf = std::max(maxf, joint_entries(c, v, true).size());
      t->scratch[c.id - 1] = nxt;
      t->maxf[c.id - 1] = (int)maxf;
      nxt += (int)maxf * conc(c);
      continue;
    }
    for (int v : c.upsched) {
      if (g->vars[v].ismargin) continue;  // never updated in the up solve (SolveTree.jl:61)
      size_t k = 0;
      for (int f : c.potentials)
        for (int q = 0; q < g->facs[f].s.nvars; q++)
          if (g->facs[f].s.vars[q] == v) { k++; break; }
      for (int chd : c.children)
        if (contains(t->cl[chd - 1].seps, v)) k++;
      maxf = std::max(maxf, k);
    }
    for (int v : c.dnsched) maxf = std::max(maxf, g->vfacs[v].size());
    t->scratch[c.id - 1] = nxt;
    t->maxf[c.id - 1] = (int)maxf;
    nxt += (int)maxf * conc(c);
  }
  t->n_slots = nxt;
  return nxt;
}
nbp_status nbp_tree_main_slots(const nbp_tree *t, int32_t *main_out, int32_t *snap_out) {
  if (!t || !main_out || t->main_slot.empty()) return hfail(NBP_ERR_ARG, "plan the slots first");
  for (size_t v = 0; v < t->main_slot.size(); v++) main_out[v] = t->main_slot[v];
  if (snap_out && !t->snap_slot.empty())
    for (size_t v = 0; v < t->snap_slot.size(); v++) snap_out[v] = t->snap_slot[v];
  return NBP_OK;
}

nbp_status nbp_tree_schedule(nbp_tree *t, uint64_t seed) {
  if (!t) return hfail(NBP_ERR_ARG, "null argument");
  if (t->main_slot.empty()) return hfail(NBP_ERR_ARG, "plan the slots first (nbp_tree_plan_slots)");
  const nbp_graph *g = t->g;
  const int n = (int)g->vars.size();
  t->stages.clear();
  t->segments.clear();
  t->seg_start = 0;
  t->meas_seed.clear();
  t->st = nbp_tree_stats{};
  std::vector<nbp_copy_desc> cps;
  if (t->snapshot) {
    for (int v = 0; v < n; v++) cps.push_back({t->snap_slot[v], t->main_slot[v]});
    add_stage(t, NBP_STAGE_COPIES, cps.data(), sizeof(nbp_copy_desc), (int)cps.size());
  }
  cps.clear();
  for (const Clique &c : t->cl)  // deep copy of the clique sub graphs (SubGraphFunctions.jl:48)
    if (t->mine(c.id))
      for (int v : c.all()) cps.push_back({t->main_slot[v], t->B[c.id - 1].at(v)});
  add_stage(t, NBP_STAGE_COPIES, cps.data(), sizeof(nbp_copy_desc), (int)cps.size());
  // heights / depths
  std::vector<int> height(t->cl.size() + 1, 0), depth(t->cl.size() + 1, 0);
  for (int cid : postorder(t)) {
    int h = 0;
    const Clique &c = t->cl[cid - 1];
    if (!c.children.empty()) {
      for (int chd : c.children) h = std::max(h, height[chd]);
      h += 1;
    }
    height[cid] = h;
  }
  int maxh = 0, maxd = 0;
  {
    std::vector<std::pair<int, int>> st;
    for (int r : t->roots) st.push_back({r, 0});
    while (!st.empty()) {
      auto [cid, d] = st.back();
      st.pop_back();
      depth[cid] = d;
      maxd = std::max(maxd, d);
      for (int chd : t->cl[cid - 1].children) st.push_back({chd, d + 1});
    }
    for (size_t k = 1; k <= t->cl.size(); k++) maxh = std::max(maxh, height[k]);
  }
  std::vector<nbp_proposal_desc> props;
  std::vector<nbp_product_desc> prods;
  nbp_status rc = NBP_OK;
  // ---- up pass: a clique starts its schedule in the stage after its last child finished (the rendezvous of the
  // CliqueStateMachine, :221-234); stage t batches step t - start[c] of every running clique (solver.TreeProgram)
  if (g->sp.upsolve) {
    const size_t nc = t->cl.size();
    std::vector<int> ids, start(nc + 1, 0), finish(nc + 1, 0);
    for (const Clique &c : t->cl) ids.push_back(c.id);
    std::stable_sort(ids.begin(), ids.end(), [&](int a, int b) { return height[a] != height[b] ? height[a] < height[b] : a < b; });
    int T = 0;
    for (int cid : ids) {
      int st = 0;
      for (int chd : t->cl[cid - 1].children) st = std::max(st, finish[chd]);
      start[cid] = st;
      finish[cid] = st + (int)t->urounds[cid - 1].size();
      T = std::max(T, finish[cid]);
    }
    for (int tt = 0; tt < T; tt++) {
      if (t->joint) {
        // prepCliqueMsgUp -> addLikelihoodsDifferentialCHILD! of the cliques that have just finished: approxDeconv
        // between the solved separator beliefs of every differential pair (searched from samples of the
        // default-constructed factor), manikde! of the result
        props.clear();
        for (const Clique &c : t->cl) {
          if (finish[c.id] != tt || c.parent == 0 || !t->mine(c.id)) continue;
          for (size_t i = 0; i < c.rel.size(); i++) {
            HFac dflt;
            memset(&dflt, 0, sizeof(dflt));
            dflt.s.factor_kind = c.rel[i][3];
            dflt.s.nvars = 2;
            dflt.s.vars[0] = c.rel[i][0];
            dflt.s.vars[1] = c.rel[i][1];
            dflt.s.ncomp = 1;
            dflt.s.comp[0][0] = 1.0;
            const int zd = dflt.s.factor_kind == NBP_F_LINREL ? mani_dim(g->vars[c.rel[i][0]].manifold) : (dflt.s.factor_kind == NBP_F_SE2 ? 3 : 1);
            for (int q = 0; q < zd; q++) dflt.s.comp[0][4 + 4 * q] = 1.0;  // identity square-root covariance
            nbp_proposal_desc d;
            fill_proposal(g, d, &dflt, -1, c.rel[i][1], &t->B[c.id - 1], &t->main_slot, nullptr, c.Dslot[i],
                          op_seed(seed, PASS_UP, c.id, 0x4000 + (int)i, 0), 0.0);
            props.push_back(d);
          }
        }
        if (!props.empty()) add_stage(t, NBP_STAGE_DECONV, props.data(), sizeof(nbp_proposal_desc), (int)props.size());
      }
      {  // up messages of the cliques that finished at tt and whose parent lives on another rank
        std::vector<int> crossing;
        for (const Clique &c : t->cl)
          if (finish[c.id] == tt && c.parent != 0 && t->own(c.id) != t->own(c.parent)) crossing.push_back(c.id);
        std::vector<Edge> edges;
        up_edges(t, crossing, edges);
        rank_exchange(t, edges);
      }
      props.clear();
      prods.clear();
      for (const Clique &c : t->cl) {
        if (!(start[c.id] <= tt && tt < finish[c.id]) || !t->mine(c.id)) continue;
        const std::vector<int> &round = t->urounds[c.id - 1][tt - start[c.id]];
        for (size_t lane = 0; lane < round.size(); lane++) {
          const int k = round[lane], v = t->usched[c.id - 1][k];
          const bool fresh = t->uiter[c.id - 1][k] == 1 || !t->stored;
          rc = update_ops(t, c.id, v, up_entries(t, c, v), nullptr, t->B[c.id - 1].at(v), PASS_UP, k, seed, props, prods, fresh, (int)lane);
          if (rc) return rc;
          t->st.updates_up++;
        }
      }
      if (!prods.empty()) {
        add_stage(t, NBP_STAGE_PROPOSALS, props.data(), sizeof(nbp_proposal_desc), (int)props.size());
        add_stage(t, NBP_STAGE_PRODUCTS, prods.data(), sizeof(nbp_product_desc), (int)prods.size());
      }
    }
  }
  if (!g->sp.downsolve) {
    cps.clear();
    for (const Clique &c : t->cl)
      if (t->mine(c.id))
        for (int v : c.frontals) cps.push_back({t->B[c.id - 1].at(v), t->main_slot[v]});
    add_stage(t, NBP_STAGE_COPIES, cps.data(), sizeof(nbp_copy_desc), (int)cps.size());
  } else {
    cps.clear();
    for (int r : t->roots)
      if (t->mine(r))
        for (int v : t->cl[r - 1].frontals) cps.push_back({t->B[r - 1].at(v), t->main_slot[v]});
    add_stage(t, NBP_STAGE_COPIES, cps.data(), sizeof(nbp_copy_desc), (int)cps.size());
    std::vector<nbp_copy_desc> final_cps;
    for (int dpt = 1; dpt <= maxd; dpt++)
      for (const Clique &c : t->cl)
        if (depth[c.id] == dpt && t->mine(c.id))
          for (int v : c.frontals) final_cps.push_back({t->B[c.id - 1].at(v), t->main_slot[v]});
    // batched by dependency like the up pass: a clique receives its parent's separator values (points-only copy)
    // and starts in the stage after the parent's last update (solver.TreeProgram._compile_down_asap)
    {
      const size_t nc = t->cl.size();
      std::vector<int> ids, start(nc + 1, 0), finish(nc + 1, 0);
      for (const Clique &c : t->cl) ids.push_back(c.id);
      std::stable_sort(ids.begin(), ids.end(), [&](int a, int b) { return depth[a] != depth[b] ? depth[a] < depth[b] : a < b; });
      int T = 0;
      for (int cid : ids) {
        const Clique &c = t->cl[cid - 1];
        start[cid] = c.parent ? finish[c.parent] : 0;
        finish[cid] = start[cid] + (int)t->drounds[cid - 1].size();
        T = std::max(T, finish[cid]);
      }
      for (int tt = 0; tt <= T; tt++) {
        // one copy stage, or one more per link of a chain of cliques without updates of their own, which hand
        // the values on within the same time step
        std::map<int, int> rnd;
        int nrounds = 0;
        for (int cid : ids) {  // depth order
          const Clique &c = t->cl[cid - 1];
          if (start[cid] != tt || !c.parent) continue;
          auto it = rnd.find(c.parent);
          rnd[cid] = it == rnd.end() ? 0 : it->second + 1;
          nrounds = std::max(nrounds, rnd[cid] + 1);
        }
        for (int r = 0; r < nrounds; r++) {
          std::vector<Edge> edges;  // down messages that cross a rank boundary: the parent's values of the child's separators
          for (const Clique &c : t->cl) {
            auto it = rnd.find(c.id);
            if (it == rnd.end() || it->second != r || t->own(c.id) == t->own(c.parent)) continue;
            for (int s : c.seps)
              edges.push_back({t->own(c.parent), t->mine(c.parent) ? t->B[c.parent - 1].at(s) : -1, t->own(c.id),
                               t->mine(c.id) ? t->B[c.id - 1].at(s) : -1});
          }
          rank_exchange(t, edges);
          cps.clear();
          for (const Clique &c : t->cl) {
            auto it = rnd.find(c.id);
            if (it == rnd.end() || it->second != r || !t->mine(c.id) || !t->mine(c.parent)) continue;
            for (int s : c.seps) cps.push_back({t->B[c.parent - 1].at(s), t->B[c.id - 1].at(s)});
          }
          if (!cps.empty()) add_stage(t, NBP_STAGE_COPY_POINTS, cps.data(), sizeof(nbp_copy_desc), (int)cps.size());  // read as points only
        }
        props.clear();
        prods.clear();
        for (const Clique &c : t->cl) {
          if (!(start[c.id] <= tt && tt < finish[c.id]) || !t->mine(c.id)) continue;
          const std::vector<int> allv = c.all();
          std::set<int> inclq(allv.begin(), allv.end());
          const std::vector<int> &round = t->drounds[c.id - 1][tt - start[c.id]];
          for (size_t lane = 0; lane < round.size(); lane++) {
            const int k = round[lane], v = t->dsched[c.id - 1][k];
            rc = update_ops(t, c.id, v, down_entries(t, c, v), &inclq, t->B[c.id - 1].at(v), PASS_DOWN, k, seed, props, prods, true, (int)lane);
            if (rc) return rc;
            t->st.updates_down++;
          }
        }
        if (!prods.empty()) {
          add_stage(t, NBP_STAGE_PROPOSALS, props.data(), sizeof(nbp_proposal_desc), (int)props.size());
          add_stage(t, NBP_STAGE_PRODUCTS, prods.data(), sizeof(nbp_product_desc), (int)prods.size());
        }
      }
    }
    // transferUpdateSubGraph!, once for the whole pass (see solver.TreeProgram)
    add_stage(t, NBP_STAGE_COPIES, final_cps.data(), sizeof(nbp_copy_desc), (int)final_cps.size());
  }
  t->segments.push_back({0, t->seg_start, (int)t->stages.size(), {}, {}});
  // statistics
  t->st.stages = (int64_t)t->stages.size();
  for (const Stage &s : t->stages) {
    if (s.kind == NBP_STAGE_PROPOSALS) t->st.proposals += s.n;
    if (s.kind == NBP_STAGE_PRODUCTS) t->st.products += s.n;
  }
  int64_t edges = 0;
  for (const Clique &c : t->cl) edges += c.parent != 0;
  t->st.messages = ((g->sp.upsolve ? 1 : 0) + (g->sp.downsolve ? 1 : 0)) * edges;
  t->st.slots = t->n_slots;
  return NBP_OK;
}

nbp_status nbp_tree_compile(nbp_tree *t, nbp_ctx *ctx, uint64_t seed, nbp_program **out) {
  if (!t || !ctx || !out) return hfail(NBP_ERR_ARG, "null argument");
  nbp_status rc = nbp_tree_schedule(t, seed);
  if (rc) return rc;
  // hand the stages to libnbp
  nbp_program *p = nullptr;
  rc = nbp_program_create(ctx, &p);
  if (rc) return rc;
  // a whole solve: the bandwidth of a belief that the same program overwrites before reading is dead
  rc = nbp_program_set_option(p, NBP_OPT_LAZY_BANDWIDTH, 1);
  if (rc) { nbp_program_destroy(p); return rc; }
  for (const Stage &s : t->stages) {
    rc = nbp_program_add_stage(p, s.kind, s.bytes.empty() ? nullptr : s.bytes.data(), s.n);
    if (rc) { nbp_program_destroy(p); return rc; }
  }
  rc = nbp_program_finalize(p);
  if (rc) { nbp_program_destroy(p); return rc; }
  *out = p;
  return NBP_OK;
}

// ---- multi-rank compile ---------------------------------------------------------------------------------------------
nbp_status nbp_tree_set_owner(nbp_tree *t, const int32_t *owner, int32_t rank) {
  if (!t) return hfail(NBP_ERR_ARG, "null argument");
  if (!owner) { t->owner.clear(); t->rank = 0; return NBP_OK; }
  if (rank < 0) return hfail(NBP_ERR_ARG, "rank < 0");
  t->owner.assign(owner, owner + t->cl.size());
  for (int o : t->owner)
    if (o < 0) return hfail(NBP_ERR_RANGE, "owner: negative rank");
  t->rank = rank;
  t->main_slot.clear();  // the slot plan depends on the ownership: plan again
  return NBP_OK;
}

// dist_solver.partition_cliques: cut the tree into >= world subtrees by repeatedly splitting the heaviest one, place the
// subtrees largest-first on the least loaded rank, give every clique above the cut to the rank of one of its children
// (level by level, siblings of a level on different ranks where possible).
// weight of a clique = 1 + the variable updates of its up schedule.
nbp_status nbp_tree_partition(const nbp_tree *t, int32_t world, int32_t *owner_out) {
  if (!t || !owner_out || world < 1) return hfail(NBP_ERR_ARG, "bad argument");
  const size_t nc = t->cl.size();
  std::vector<double> w(nc + 1, 0.0), sub(nc + 1, 0.0);
  for (const Clique &c : t->cl) w[c.id] = 1.0 + (double)c.upsched.size();
  for (int cid : postorder(t)) {
    sub[cid] = w[cid];
    for (int chd : t->cl[cid - 1].children) sub[cid] += sub[chd];
  }
  double total = 0;
  for (size_t k = 1; k <= nc; k++) total += w[k];
  std::vector<int> roots(t->roots.begin(), t->roots.end()), top;
  while ((int)roots.size() < 6 * world) {
    int best = -1;
    for (int r : roots)
      if (!t->cl[r - 1].children.empty() && (best < 0 || sub[r] > sub[best])) best = r;  // first maximum, like Python's max()
    if (best < 0) break;
    if ((int)roots.size() >= world && sub[best] <= 0.6 * total / world) break;
    roots.erase(std::find(roots.begin(), roots.end(), best));
    top.push_back(best);
    for (int chd : t->cl[best - 1].children) roots.push_back(chd);
  }
  std::vector<int> owner(nc + 1, 0);
  std::vector<double> load(world, 0.0);
  std::stable_sort(roots.begin(), roots.end(), [&](int a, int b) { return sub[a] > sub[b]; });
  for (int r : roots) {
    int k = 0;
    for (int q = 1; q < world; q++)
      if (load[q] < load[k]) k = q;  // argmin: first minimum
    load[k] += sub[r];
    std::vector<int> st{r};
    while (!st.empty()) {
      const int c = st.back();
      st.pop_back();
      owner[c] = k;
      for (int chd : t->cl[c - 1].children) st.push_back(chd);
    }
  }
  // The cliques above the cut, level by level (a clique's level = the longest chain of such cliques below it): each goes
  // to the rank of one of its children -- one of its edges stays local -- and the cliques of one level, which can run side
  // by side, go to different ranks where their children allow it: the heaviest child whose rank has no clique of this
  // level yet, else the heaviest child.
  {
    std::map<int, int> level;
    int maxlevel = 0;
    for (auto it = top.rbegin(); it != top.rend(); ++it) {  // children before parents
      int lv = 0;
      for (int chd : t->cl[*it - 1].children)
        if (level.count(chd)) lv = std::max(lv, level[chd] + 1);
      level[*it] = lv;
      maxlevel = std::max(maxlevel, lv);
    }
    for (int lv = 0; lv <= maxlevel; lv++) {
      std::vector<char> used(world, 0);
      for (auto it = top.rbegin(); it != top.rend(); ++it) {
        if (level[*it] != lv) continue;
        std::vector<int> kids(t->cl[*it - 1].children.begin(), t->cl[*it - 1].children.end());
        std::stable_sort(kids.begin(), kids.end(), [&](int a, int b) { return sub[a] > sub[b]; });
        int pick = owner[kids[0]];
        for (int chd : kids)
          if (!used[owner[chd]]) { pick = owner[chd]; break; }
        owner[*it] = pick;
        used[pick] = 1;
        load[pick] += w[*it];
      }
    }
  }
  for (size_t k = 1; k <= nc; k++) owner_out[k - 1] = owner[k];
  return NBP_OK;
}

// One solve of this rank's share: the stage segments of the last compile with the separator exchanges between them, all
// issued from here on the library's stream -- no host code between a segment and the exchange behind it, no host
// synchronisation (the reference: one Task per clique blocking on its Channels, SolverAPI.jl:50-100).
nbp_status nbp_tree_run_sharded_cb(const nbp_tree *t, nbp_program *prog, nbp_exchange_fn xchg, void *user) {
  if (!t || !prog) return hfail(NBP_ERR_ARG, "null argument");
  std::vector<nbp_xfer> sx, rx;
  for (const nbp_tree::Segment &x : t->segments) {
    if (x.kind == 0) {
      if (x.last > x.first) {
        nbp_status rc = nbp_program_run(prog, x.first, x.last);
        if (rc) return rc;
      }
      continue;
    }
    if (x.sends.empty() && x.recvs.empty()) continue;
    if (!xchg) return hfail(NBP_ERR_ARG, "run_sharded: the compile has exchange segments but no transport was given");
    sx.clear();
    rx.clear();
    for (const auto &s : x.sends) sx.push_back({s[0], s[1]});
    for (const auto &r : x.recvs) rx.push_back({r[0], r[1]});
    nbp_status rc = xchg(user, sx.data(), (int32_t)sx.size(), rx.data(), (int32_t)rx.size());
    if (rc) return rc;
  }
  return NBP_OK;
}
namespace {
struct rccl_transport { nbp_ctx *ctx; nbp_comm *comm; };
nbp_status rccl_xchg(void *user, const nbp_xfer *sends, int32_t ns, const nbp_xfer *recvs, int32_t nr) {
  rccl_transport *r = (rccl_transport *)user;
  return nbp_exchange(r->ctx, r->comm, sends, ns, recvs, nr);
}
}  // namespace
nbp_status nbp_tree_run_sharded(const nbp_tree *t, nbp_program *prog, nbp_ctx *ctx, nbp_comm *comm) {
  if (!t || !prog) return hfail(NBP_ERR_ARG, "null argument");
  rccl_transport r{ctx, comm};
  return nbp_tree_run_sharded_cb(t, prog, (ctx && comm) ? rccl_xchg : nullptr, &r);
}

int32_t nbp_tree_num_segments(const nbp_tree *t) { return t ? (int32_t)t->segments.size() : 0; }
nbp_status nbp_tree_segment(const nbp_tree *t, int32_t i, int32_t *kind, int32_t *first, int32_t *last, int32_t *nsend, int32_t *nrecv,
                            nbp_xfer *sends, nbp_xfer *recvs, int32_t cap) {
  if (!t || i < 0 || i >= (int)t->segments.size()) return hfail(NBP_ERR_RANGE, "segment index");
  const nbp_tree::Segment &x = t->segments[i];
  if (kind) *kind = x.kind;
  if (first) *first = x.first;
  if (last) *last = x.last;
  if (nsend) *nsend = (int32_t)x.sends.size();
  if (nrecv) *nrecv = (int32_t)x.recvs.size();
  for (size_t k = 0; sends && k < x.sends.size() && (int)k < cap; k++) sends[k] = {x.sends[k][0], x.sends[k][1]};
  for (size_t k = 0; recvs && k < x.recvs.size() && (int)k < cap; k++) recvs[k] = {x.recvs[k][0], x.recvs[k][1]};
  return NBP_OK;
}

nbp_status nbp_tree_get_stats(const nbp_tree *t, nbp_tree_stats *out) {
  if (!t || !out) return hfail(NBP_ERR_ARG, "null argument");
  *out = t->st;
  return NBP_OK;
}
// ---- graph initialisation: initAll! / doautoinit! (GraphInit.jl:61-199), solver.initStages ---------
int32_t nbp_graph_init_plan(nbp_graph *g, uint64_t seed) {
  if (!g) return hfail(NBP_ERR_ARG, "null argument");
  const int V = (int)g->vars.size();
  std::vector<char> init(V);
  for (int v = 0; v < V; v++) init[v] = g->vars[v].initialized;
  struct PlanItem { int sym; std::vector<int> use; std::vector<char> state; };
  std::vector<PlanItem> plan;
  for (int sweep = 0; sweep < 10; sweep++) {
    bool repeat = false;
    for (int sym = 0; sym < V; sym++) {
      if (init[sym]) continue;
      std::vector<int> use;
      for (int fl : g->vfacs[sym]) {
        const nbp_factor_spec &f = g->facs[fl].s;
        bool ok = true;  // priors and general n-ary cases: everything else initialised (GraphInit.jl:90)
        for (int k = 0; k < f.nvars; k++)
          if (f.vars[k] != sym && !init[f.vars[k]]) ok = false;
        if (!ok && f.has_multihypo) {  // at least one hypothesis available (:94-105, FactorGraph.jl:772-784)
          bool sym_cer = false, sym_unc = false, any_unc = false, all_cer = true;
          for (int k = 0; k < f.nvars; k++) {
            const bool cer = f.multihypo[k] == 0.0, unc = f.multihypo[k] > 0.0;
            if (f.vars[k] == sym) { sym_cer |= cer; sym_unc |= unc; }
            if (unc && init[f.vars[k]]) any_unc = true;
            if (cer && !init[f.vars[k]]) all_cer = false;
          }
          ok = (sym_cer && any_unc) || (sym_unc && all_cer);
        }
        if (ok) use.push_back(fl);
      }
      if (!use.empty()) {
        plan.push_back({sym, use, init});
        init[sym] = 1;
      } else
        repeat = true;
    }
    if (!repeat) break;
  }
  g->init_vars.clear();
  g->init_stages.clear();
  g->init_slots = V;
  if (plan.empty()) return V;
  size_t maxF = 0;
  for (auto &p : plan) maxF = std::max(maxF, p.use.size());
  if (maxF > NBP_MAXF) return hfail(NBP_ERR_RANGE, "graph init: a product exceeds NBP_MAXF densities");
  std::vector<std::vector<const PlanItem *>> groups(1);
  std::set<int> produced;
  for (auto &p : plan) {
    bool dep = false;
    for (int fl : p.use)
      for (int k = 0; k < g->facs[fl].s.nvars; k++)
        if (g->facs[fl].s.vars[k] != p.sym && produced.count(g->facs[fl].s.vars[k])) dep = true;
    if (dep) {
      groups.emplace_back();
      produced.clear();
    }
    groups.back().push_back(&p);
    produced.insert(p.sym);
  }
  size_t width = 0;
  for (auto &gr : groups) width = std::max(width, gr.size());
  g->init_dens0 = V + (int)(width * maxF);  // V beliefs | proposal scratch | the pass-through densities
  nbp_tree tmp;  // only for add_stage's container
  tmp.g = g;
  for (auto &gr : groups) {
    std::vector<nbp_proposal_desc> props;
    std::vector<nbp_product_desc> prods;
    for (size_t ci = 0; ci < gr.size(); ci++) {
      const PlanItem &p = *gr[ci];
      bool anymh = false, anypartial = false;
      for (int fl : p.use) anymh |= g->facs[fl].s.has_multihypo != 0;
      const int base = V + (int)(ci * maxF);
      nbp_product_desc q;
      memset(&q, 0, sizeof(q));
      for (size_t i = 0; i < p.use.size(); i++) {
        const HFac &fac = g->facs[p.use[i]];
        const double ns = (anymh && !fac.is_prior && !fac.s.has_multihypo) ? g->sp.null_surplus_add : 0.0;
        nbp_proposal_desc d;
        fill_proposal(g, d, &fac, fac.dens >= 0 ? g->init_dens0 + fac.dens : -1, p.sym, nullptr, nullptr, nullptr, base + (int)i,
                      op_seed(seed, PASS_INIT, p.sym, 0, i + 1), ns, &p.state, p.use.size() == 1 ? 2 : 0);
        props.push_back(d);
        q.in_slot[i] = base + (int)i;
        q.in_partial[i] = (uint8_t)fac.s.partial_mask;
        anypartial |= fac.s.partial_mask != 0;
      }
      q.manifold = g->vars[p.sym].manifold;
      q.nfactors = (int)p.use.size();
      q.niter = g->sp.product_niter;
      q.out_slot = p.sym;
      q.labels_out = -1;
      q.old_slot = anypartial ? p.sym : -1;
      if (!anypartial) memset(q.in_partial, 0, sizeof(q.in_partial));
      q.seed = op_seed(seed, PASS_INIT, p.sym, 0, PRODUCT_ID);
      prods.push_back(q);
    }
    add_stage(&tmp, NBP_STAGE_PROPOSALS, props.data(), sizeof(nbp_proposal_desc), (int)props.size());
    add_stage(&tmp, NBP_STAGE_PRODUCTS, prods.data(), sizeof(nbp_product_desc), (int)prods.size());
  }
  g->init_stages = std::move(tmp.stages);
  for (auto &p : plan) g->init_vars.push_back(p.sym);
  g->init_slots = V + (int)(width * maxF) + (int)g->dens_facs.size();
  return g->init_slots;
}

// pass-through priors: factor ids in density order, and where their slots start in the two slot plans
int32_t nbp_graph_num_densities(const nbp_graph *g) { return g ? (int32_t)g->dens_facs.size() : 0; }
nbp_status nbp_graph_density_factors(const nbp_graph *g, int32_t *out) {
  if (!g || !out) return hfail(NBP_ERR_ARG, "null argument");
  for (size_t i = 0; i < g->dens_facs.size(); i++) out[i] = g->dens_facs[i];
  return NBP_OK;
}
int32_t nbp_graph_init_density_slot0(const nbp_graph *g) { return g ? g->init_dens0 : hfail(NBP_ERR_ARG, "null argument"); }
int32_t nbp_tree_density_slot0(const nbp_tree *t) { return t ? t->dens0 : hfail(NBP_ERR_ARG, "null argument"); }

int32_t nbp_graph_init_num_variables(const nbp_graph *g) { return g ? (int32_t)g->init_vars.size() : 0; }
nbp_status nbp_graph_init_variables(const nbp_graph *g, int32_t *out) {
  if (!g || !out) return hfail(NBP_ERR_ARG, "null argument");
  for (size_t i = 0; i < g->init_vars.size(); i++) out[i] = g->init_vars[i];
  return NBP_OK;
}
int32_t nbp_graph_init_num_stages(const nbp_graph *g) { return g ? (int32_t)g->init_stages.size() : 0; }
nbp_status nbp_graph_init_stage(const nbp_graph *g, int32_t s, int32_t *kind, int32_t *n, void *out, int64_t cap) {
  if (!g || s < 0 || s >= (int)g->init_stages.size()) return hfail(NBP_ERR_RANGE, "stage index");
  const Stage &st = g->init_stages[s];
  if (kind) *kind = st.kind;
  if (n) *n = st.n;
  if (out && cap > 0) memcpy(out, st.bytes.data(), std::min<size_t>((size_t)cap, st.bytes.size()));
  return NBP_OK;
}
nbp_status nbp_graph_init_compile(nbp_graph *g, nbp_ctx *ctx, nbp_program **out) {
  if (!g || !ctx || !out) return hfail(NBP_ERR_ARG, "null argument");
  nbp_program *p = nullptr;
  nbp_status rc = nbp_program_create(ctx, &p);
  if (rc) return rc;
  for (const Stage &s : g->init_stages) {
    rc = nbp_program_add_stage(p, s.kind, s.bytes.empty() ? nullptr : s.bytes.data(), s.n);
    if (rc) { nbp_program_destroy(p); return rc; }
  }
  rc = nbp_program_finalize(p);
  if (rc) { nbp_program_destroy(p); return rc; }
  for (int v : g->init_vars) g->vars[v].initialized = true;  // what the program does when it is run
  *out = p;
  return NBP_OK;
}

// ---- the clique seam, one clique at a time (upGibbsCliqueDensity / solveCliqDownFrontalProducts!) ---------
static nbp_status clique_check(const nbp_solver_params *sp, const nbp_clique_desc *q) {
  if (!sp || !q) return hfail(NBP_ERR_ARG, "null argument");
  if (q->n_diff < 0 || (q->n_diff > 0 && (!q->diff_a || !q->diff_b || !q->diff_kind))) return hfail(NBP_ERR_ARG, "clique: differential lists");
  for (int i = 0; i < q->n_diff; i++) {
    if (q->diff_a[i] < 0 || q->diff_a[i] >= q->nvars || q->diff_b[i] < 0 || q->diff_b[i] >= q->nvars || q->diff_a[i] == q->diff_b[i])
      return hfail(NBP_ERR_RANGE, "clique: differential variable index");
    if (q->diff_kind[i] != NBP_F_LINREL && q->diff_kind[i] != NBP_F_CIRCULAR && q->diff_kind[i] != NBP_F_SE2)
      return hfail(NBP_ERR_ARG, "clique: a differential factor is LinearRelative, CircularCircular or an SE(2) ManifoldFactor");
  }
  if (q->nvars < 1 || q->nfrontals < 1 || q->nseparators < 0 || q->nfrontals + q->nseparators > q->nvars)
    return hfail(NBP_ERR_RANGE, "clique: variable counts");
  if (!q->manifold || (q->nfactors > 0 && !q->factors) || (q->nmsgs > 0 && (!q->msg_var || !q->msg_belief)))
    return hfail(NBP_ERR_ARG, "clique: null array");
  for (int v = 0; v < q->nvars; v++)
    if (q->manifold[v] < NBP_EUCLID1 || q->manifold[v] > NBP_SE2) return hfail(NBP_ERR_ARG, "clique: unknown manifold");
  for (int f = 0; f < q->nfactors; f++) {
    const nbp_factor_spec &s = q->factors[f];
    if (s.factor_kind < NBP_F_PRIOR || s.factor_kind > NBP_F_PASSTHROUGH || s.factor_kind == NBP_F_MSGPRIOR) return hfail(NBP_ERR_ARG, "clique: factor kind");
    if (s.factor_kind == NBP_F_PASSTHROUGH) {
      if (s.nvars != 1 || s.has_multihypo) return hfail(NBP_ERR_ARG, "clique: a pass-through prior is unary, without multihypo");
      if (!q->factor_density || !q->factor_density[f].pts || !q->factor_density[f].bw)
        return hfail(NBP_ERR_ARG, "clique: a pass-through prior needs its density (factor_density[f])");
    }
    if (s.nvars < 1 || s.nvars > NBP_MAXV || s.ncomp < 1 || s.ncomp > NBP_MAXC) return hfail(NBP_ERR_RANGE, "clique: factor shape");
    if (spec_has_table(s)) {
      if (s.nvars >= NBP_MAXV) return hfail(NBP_ERR_RANGE, "clique: a factor with a sampler table takes at most NBP_MAXV - 1 variables");
      if (!q->factor_density || !q->factor_density[f].pts || !q->factor_density[f].bw || q->factor_density[f].n_pts < 1)
        return hfail(NBP_ERR_ARG, "clique: an AliasingScalarSampler measurement needs its table (factor_density[f])");
      // the table lives in a belief slot (N rows): a longer one would be cut to its first N entries and the tail's mass would
      // land on entry N - 1 -- refused (the reference takes any length, entities/AliasScalarSampling.jl:13-55: such a clique
      // goes down the caller's generic path, ext/IIFNbpExt.jl `supported`)
      if (q->factor_density[f].n_pts > sp->N)
        return hfail(NBP_ERR_RANGE, "clique: an AliasingScalarSampler table holds at most N entries (it lives in a belief slot)");
    }
    for (int i = 0; i < s.nvars; i++)
      if (s.vars[i] < 0 || s.vars[i] >= q->nvars) return hfail(NBP_ERR_RANGE, "clique: factor variable index");
  }
  for (int i = 0; i < q->nmsgs; i++)
    if (q->msg_var[i] < 0 || q->msg_var[i] >= q->nvars) return hfail(NBP_ERR_RANGE, "clique: message variable index");
  auto chk = [&](const int32_t *l, int n) {
    if (n < 0 || (n > 0 && !l)) return false;
    for (int i = 0; i < n; i++)
      if (l[i] < 0 || l[i] >= q->nvars) return false;
    return true;
  };
  if (!chk(q->direct_frtl_msg, q->n_direct_frtl_msg) || !chk(q->msgskip, q->n_msgskip) || !chk(q->itervar, q->n_itervar) ||
      !chk(q->direct_prior_msg, q->n_direct_prior_msg))
    return hfail(NBP_ERR_RANGE, "clique: Gibbs id list");
  return NBP_OK;
}

// the densities of variable v: the factors that touch it (in the caller's order), then the messages on it
static void clique_entries(const nbp_clique_desc *q, int v, bool with_msgs, std::vector<int> &facs, std::vector<int> &msgs) {
  facs.clear();
  msgs.clear();
  for (int f = 0; f < q->nfactors; f++)
    for (int i = 0; i < q->factors[f].nvars; i++)
      if (q->factors[f].vars[i] == v) { facs.push_back(f); break; }
  if (with_msgs)
    for (int i = 0; i < q->nmsgs; i++)
      if (q->msg_var[i] == v) msgs.push_back(i);
}

int32_t nbp_clique_slots(const nbp_clique_desc *q) {
  if (!q) return hfail(NBP_ERR_ARG, "null argument");
  size_t maxf = 1;
  std::vector<int> fa, ms;
  for (int v = 0; v < q->nvars; v++) {
    clique_entries(q, v, true, fa, ms);
    maxf = std::max(maxf, fa.size() + ms.size());
  }
  int ndens = 0, nkde = 0;
  for (int f = 0; f < q->nfactors; f++) {
    ndens += q->factors && (q->factors[f].factor_kind == NBP_F_PASSTHROUGH || spec_has_table(q->factors[f]));
    nkde += q->factor_meas_kde && q->factor_meas_kde[f].pts != nullptr;
  }
  // steps that commute run side by side, one scratch row of maxf proposals each: at most one per variable
  return q->nvars + q->nmsgs + ndens + nkde + (q->n_diff > 0 ? q->n_diff : 0) + (int32_t)maxf * q->nvars;
}
static size_t clique_maxf(const nbp_clique_desc *q) {
  size_t maxf = 1;
  std::vector<int> fa, ms;
  for (int v = 0; v < q->nvars; v++) {
    clique_entries(q, v, true, fa, ms);
    maxf = std::max(maxf, fa.size() + ms.size());
  }
  return maxf;
}

// measurement dimension of a relative factor kind on a variable's manifold
static int clique_zdim(int kind, int manifold) { return kind == NBP_F_LINREL ? mani_dim(manifold) : (kind == NBP_F_SE2 ? 3 : 1); }

// One clique call, planned: the beliefs to move in, the rounds of (proposals, products), the differential stage, the
// beliefs to move out -- with every slot number shifted by `off`, so that several cliques can share one context, one
// transfer each way and one program whose stage pair r holds round r of every clique (nbp_clique_solve_batch).
// host-side wall clock of the phases of the clique calls (diagnostics: nbp_clique_seam_times): 0 planning, 1 beliefs in,
// 2 program assembly + finalize, 3 launches + waiting for them, 4 beliefs out, 5 calls
// (atomics: the reference runs cliques as concurrent tasks, and clique calls on different contexts may come from different
//  host threads at once -- plain globals updated by all of them were a data race)
static std::atomic<double> g_seam_s[6];
static std::atomic<bool> g_seam_sync{false};
static inline void seam_add(int i, double v) {
  double cur = g_seam_s[i].load(std::memory_order_relaxed);
  while (!g_seam_s[i].compare_exchange_weak(cur, cur + v, std::memory_order_relaxed)) {}
}
static inline double seam_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct clique_io { int32_t slot, mani; const nbp_tree_belief *src; nbp_tree_belief *dst; bool with_ipc; };
struct clique_plan {
  int nslots = 0;  // slots this clique occupies: [0, base) beliefs that cross the boundary, [base, nslots) proposal scratch
  int base = 0;
  std::vector<clique_io> in, out;
  std::vector<std::pair<std::vector<nbp_proposal_desc>, std::vector<nbp_product_desc>>> rounds;
  std::vector<nbp_proposal_desc> deconv;
};
// Several cliques in one context: the beliefs that cross the boundary (variables, messages, densities, measurement KDEs,
// differentials: local slots [0, base)) of ALL cliques lie one behind the other from offA on, the scratch rows of all of
// them behind that (from offB on) -- so the beliefs of a whole batch travel in one copy each way.
struct slot_map {
  int base, offA, offB;
  int operator()(int s) const { return s < base ? s + offA : s - base + offB; }
};
static void relocate(nbp_proposal_desc &d, const slot_map &m) {
  for (int i = 0; i < NBP_MAXV; i++) d.var_slot[i] = m(d.var_slot[i]);  // (entries beyond nvars are never read)
  d.out_slot = m(d.out_slot);
  if (d.meas_kde > 0) d.meas_kde = m(d.meas_kde - 1) + 1;
}
static void relocate(nbp_product_desc &d, const slot_map &m) {
  for (int i = 0; i < d.nfactors; i++) d.in_slot[i] = m(d.in_slot[i]);
  d.out_slot = m(d.out_slot);
  if (d.old_slot >= 0) d.old_slot = m(d.old_slot);
}
static nbp_status clique_plan_build(const nbp_solver_params *sp, const nbp_clique_desc *q, uint64_t seed, nbp_tree_belief *bel, bool down,
                                    nbp_tree_belief *diff_out, clique_plan &P) {
  nbp_status rc = clique_check(sp, q);
  if (rc) return rc;
  if (!bel) return hfail(NBP_ERR_ARG, "null argument");
  if (!down && q->n_diff > 0 && !diff_out) return hfail(NBP_ERR_ARG, "clique: differential factors are asked for, diff_out is null (nbp_clique_upsolve_joint)");
  // a throw-away graph object carries the solver parameters and the variables for fill_proposal
  nbp_graph g;
  g.sp = *sp;
  for (int v = 0; v < q->nvars; v++) g.vars.push_back({q->manifold[v], true, q->ismargin ? q->ismargin[v] != 0 : false});
  std::vector<HFac> facs(q->nfactors);
  int ndens = 0;
  for (int f = 0; f < q->nfactors; f++) {
    facs[f].s = q->factors[f];
    facs[f].is_prior = q->factors[f].factor_kind == NBP_F_PRIOR || q->factors[f].factor_kind == NBP_F_PASSTHROUGH;
    if (q->factors[f].factor_kind == NBP_F_PASSTHROUGH || spec_has_table(q->factors[f])) facs[f].dens = ndens++;
  }
  // slot plan: the clique's variables | the message beliefs | the pass-through densities | the measurement KDEs of
  // differential factors received from children | the differential KDEs this clique sends up | proposal scratch
  std::vector<int> kde_of(q->nfactors, -1);
  int nkde = 0;
  for (int f = 0; f < q->nfactors; f++)
    if (q->factor_meas_kde && q->factor_meas_kde[f].pts) {
      const int k = q->factors[f].factor_kind;
      if ((k != NBP_F_LINREL && k != NBP_F_CIRCULAR && k != NBP_F_SE2) || q->factors[f].nvars != 2 || !q->factor_meas_kde[f].bw)
        return hfail(NBP_ERR_ARG, "clique: a measurement KDE (points + bandwidth) belongs to a binary LinearRelative / CircularCircular / SE(2) factor");
      kde_of[f] = nkde++;
    }
  const int ndiff = down ? 0 : std::max(0, (int)q->n_diff);
  const int msg0 = q->nvars, dens0 = q->nvars + q->nmsgs, kde0 = dens0 + ndens, diff0 = kde0 + nkde, base = diff0 + ndiff;
  // ---- schedule ------------------------------------------------------------------------------------------
  std::vector<int> sched, iter;
  std::vector<int> fa, ms;
  if (!down) {  // upGibbsCliqueDensity: four fmcmc! passes (SolveTree.jl:193-235); one label -> one iteration (:106-108)
    auto fmcmc = [&](const std::vector<int> &l, int it) {
      if (l.size() == 1) it = 1;
      for (int k = 0; k < it; k++)
        for (int v : l) { sched.push_back(v); iter.push_back(k + 1); }
    };
    auto vec = [](const int32_t *l, int n) { return std::vector<int>(l, l + n); };
    const std::vector<int> skip = vec(q->msgskip, q->n_msgskip);
    fmcmc(vec(q->direct_frtl_msg, q->n_direct_frtl_msg), 1);
    if (!skip.empty()) fmcmc(skip, 1);
    if (q->n_itervar > 0) fmcmc(vec(q->itervar, q->n_itervar), sp->gibbs_iters);
    if (q->n_direct_prior_msg > 0) {
      std::vector<int> l;
      for (int i = 0; i < q->n_direct_prior_msg; i++)
        if (!contains(skip, (int)q->direct_prior_msg[i])) l.push_back(q->direct_prior_msg[i]);
      fmcmc(l, 1);
    }
    // doFMCIteration: marginalized variables and variables without any density are passed over (:61)
    std::vector<int> s2, i2;
    for (size_t k = 0; k < sched.size(); k++) {
      clique_entries(q, sched[k], true, fa, ms);
      if ((fa.empty() && ms.empty()) || g.vars[sched[k]].ismargin) continue;
      s2.push_back(sched[k]);
      i2.push_back(iter[k]);
    }
    sched.swap(s2);
    iter.swap(i2);
  } else {  // determineCliqVariableDownSequence + solveCliqDownFrontalProducts! (CliqStateMachineUtils.jl:424-571)
    std::vector<int> iterv;
    for (int f = 0; f < q->nfactors; f++) {
      std::vector<int> hit;
      for (int i = 0; i < q->factors[f].nvars; i++)
        if (q->factors[f].vars[i] < q->nfrontals) hit.push_back(q->factors[f].vars[i]);
      if (hit.size() > 1)
        for (int v : hit)
          if (!contains(iterv, v)) iterv.push_back(v);
    }
    std::vector<int> itf, directs;
    for (int v = 0; v < q->nfrontals; v++) {
      if (sp->limitfixeddown && g.vars[v].ismargin) continue;
      (contains(iterv, v) ? itf : directs).push_back(v);
    }
    for (int v : directs) { sched.push_back(v); iter.push_back(1); }
    for (int k = 0; k < NBP_DOWN_MCITERS; k++)
      for (int v : itf) { sched.push_back(v); iter.push_back(k + 1); }
  }
  // ---- beliefs in --------------------------------------------------------------------------------------------------
  {
    auto put = [&](int slot, int mani, const nbp_tree_belief &m, bool with_ipc) { P.in.push_back({slot, mani, &m, nullptr, with_ipc}); };
    for (int v = 0; v < q->nvars; v++) {
      if (!bel[v].pts && bel[v].handle <= 0) return hfail(NBP_ERR_ARG, "clique: null belief");
      put(v, q->manifold[v], bel[v], true);
    }
    for (int i = 0; i < (down ? 0 : q->nmsgs); i++) {
      const nbp_tree_belief &m = q->msg_belief[i];
      if ((!m.pts || !m.bw) && m.handle <= 0) return hfail(NBP_ERR_ARG, "clique: a message needs points and bandwidth");
      put(msg0 + i, q->manifold[q->msg_var[i]], m, true);
    }
    for (int f = 0; f < q->nfactors; f++)
      if (facs[f].dens >= 0)  // a density in the variable's layout, or a sampler table (two rows: domain, cumulative weights)
        put(dens0 + facs[f].dens, facs[f].s.factor_kind == NBP_F_PASSTHROUGH ? q->manifold[facs[f].s.vars[0]] : (int32_t)NBP_EUCLID2,
            q->factor_density[f], true);
    for (int f = 0; f < q->nfactors; f++)  // LinearRelative(::MKD) & co.: the measurement is the child's KDE, in measurement coordinates
      if (kde_of[f] >= 0) put(kde0 + kde_of[f], clique_zdim(q->factors[f].factor_kind, q->manifold[q->factors[f].vars[0]]) /* Euclid(zd) */, q->factor_meas_kde[f], false);
  }
  // ---- the schedule -------------------------------------------------------------------------------------------------
  const bool stored = (sp->flags & NBP_SOLVER_STORED_MEASUREMENTS) != 0;
  std::map<std::pair<int, int>, uint64_t> meas_seed;  // (0 = factor | 1 = message, index) -> seed of its last fresh draw
  const int passid = down ? PASS_DOWN : PASS_UP;
  std::vector<char> updated(q->nvars, 0);
  // rounds of commuting steps (solver.TreeProgram._rounds): steps whose variables differ and share no factor read and
  // write disjoint beliefs, and the random streams are keyed by the step index -- one stage pair per round
  std::vector<std::vector<int>> rounds;
  {
    std::vector<std::set<int>> reads(q->nvars);
    for (int v = 0; v < q->nvars; v++) {
      clique_entries(q, v, false, fa, ms);
      for (int f : fa)
        for (int i = 0; i < q->factors[f].nvars; i++)
          if (q->factors[f].vars[i] != v) reads[v].insert(q->factors[f].vars[i]);
    }
    std::vector<int> rnd(sched.size(), 0);
    for (size_t j = 0; j < sched.size(); j++) {
      for (size_t i = 0; i < j; i++)
        if (sched[i] == sched[j] || reads[sched[j]].count(sched[i]) || reads[sched[i]].count(sched[j])) rnd[j] = std::max(rnd[j], rnd[i] + 1);
      if ((int)rounds.size() <= rnd[j]) rounds.resize(rnd[j] + 1);
      rounds[rnd[j]].push_back((int)j);
    }
  }
  const int maxf = (int)clique_maxf(q);
  for (const std::vector<int> &round : rounds) {
    std::vector<nbp_proposal_desc> props;
    std::vector<nbp_product_desc> prods;
    for (size_t lane = 0; lane < round.size(); lane++) {
      const size_t k = (size_t)round[lane];
      const int v = sched[k];
      clique_entries(q, v, !down, fa, ms);
      const int F = (int)(fa.size() + ms.size());
      if (F == 0) continue;
      if (F > NBP_MAXF) return hfail(NBP_ERR_RANGE, "a product exceeds NBP_MAXF densities");
      bool anymh = false;
      for (int f : fa) anymh |= facs[f].s.has_multihypo != 0;
      nbp_product_desc pq;
      memset(&pq, 0, sizeof(pq));
      bool anypartial = false;
      const bool fresh = iter[k] == 1 || !stored || down;
      const int row = base + (int)lane * maxf;
      for (int i = 0; i < F; i++) {
        const bool ismsg = i >= (int)fa.size();
        const HFac *fac = ismsg ? nullptr : &facs[fa[i]];
        const int mi = ismsg ? ms[i - fa.size()] : -1;
        double ns = 0.0;  // proposalbeliefs!: relative non-multihypo siblings of a multihypo factor (ApproxConv.jl:255-265)
        if (anymh && fac && !fac->is_prior && !fac->s.has_multihypo) ns = sp->null_surplus_add;
        nbp_proposal_desc d;
        const uint64_t sd = op_seed(seed, passid, q->clique_id, (uint64_t)k, (uint64_t)(i + 1));
        fill_proposal(&g, d, fac, ismsg ? msg0 + mi : (fac->dens >= 0 ? dens0 + fac->dens : -1), v, nullptr, nullptr, nullptr, row + i, sd, ns,
                      nullptr, F == 1 ? 1 : 0);
        if (!ismsg && kde_of[fa[i]] >= 0) d.meas_kde = kde0 + kde_of[fa[i]] + 1;
        const std::pair<int, int> key{ismsg ? 1 : 0, ismsg ? mi : fa[i]};
        if (fresh) meas_seed[key] = sd;
        else {
          auto it = meas_seed.find(key);
          d.meas_seed = it == meas_seed.end() ? 0 : it->second;
        }
        props.push_back(d);
        pq.in_slot[i] = row + i;
        pq.in_partial[i] = (uint8_t)(fac ? fac->s.partial_mask : 0);
        anypartial |= pq.in_partial[i] != 0;
      }
      pq.manifold = q->manifold[v];
      pq.nfactors = F;
      pq.niter = sp->product_niter;
      pq.out_slot = v;
      pq.labels_out = -1;
      pq.old_slot = anypartial ? v : -1;
      if (!anypartial) memset(pq.in_partial, 0, sizeof(pq.in_partial));
      pq.seed = op_seed(seed, passid, q->clique_id, (uint64_t)k, PRODUCT_ID);
      prods.push_back(pq);
      updated[v] = 1;
    }
    if (prods.empty()) continue;
    P.rounds.emplace_back(std::move(props), std::move(prods));
  }
  int lanes = 1;
  for (const std::vector<int> &round : rounds) lanes = std::max(lanes, (int)round.size());
  P.base = base;
  P.nslots = base + lanes * maxf;
  if (ndiff > 0) {
    // prepCliqueMsgUp -> addLikelihoodsDifferentialCHILD! (TreeMessageUtils.jl:279-335): approxDeconv between the solved
    // beliefs of every pair, searched from samples of the default-constructed factor, manikde! of the result -- the same
    // ops, with the same seeds, as the whole-tree compile emits when the clique finishes
    std::vector<nbp_proposal_desc> props;
    for (int i = 0; i < ndiff; i++) {
      HFac dflt;
      memset(&dflt, 0, sizeof(dflt));
      dflt.s.factor_kind = q->diff_kind[i];
      dflt.s.nvars = 2;
      dflt.s.vars[0] = q->diff_a[i];
      dflt.s.vars[1] = q->diff_b[i];
      dflt.s.ncomp = 1;
      dflt.s.comp[0][0] = 1.0;
      const int zd = clique_zdim(q->diff_kind[i], q->manifold[q->diff_a[i]]);
      for (int k = 0; k < zd; k++) dflt.s.comp[0][4 + 4 * k] = 1.0;  // identity square-root covariance
      nbp_proposal_desc d;
      fill_proposal(&g, d, &dflt, -1, q->diff_b[i], nullptr, nullptr, nullptr, diff0 + i, op_seed(seed, PASS_UP, q->clique_id, 0x4000 + i, 0), 0.0);
      props.push_back(d);
    }
    P.deconv = std::move(props);
  }
  // ---- beliefs out: setValKDE!(vnd, mkd, setinit, ipc) (FactorGraph.jl:250-263) for everything the schedule touched, then
  // the differential KDEs (points in measurement coordinates + fitted bandwidth)
  for (int v = 0; v < q->nvars; v++)
    if (updated[v]) P.out.push_back({v, q->manifold[v], nullptr, &bel[v], true});
  for (int i = 0; i < ndiff; i++) {
    if (!diff_out[i].pts || !diff_out[i].bw) return hfail(NBP_ERR_ARG, "clique: diff_out entries need pts and bw");
    P.out.push_back({diff0 + i, clique_zdim(q->diff_kind[i], q->manifold[q->diff_a[i]]), nullptr, &diff_out[i], false});
  }
  return NBP_OK;
}

// the plans of one or several cliques on one context: one transfer in, one program (stage pair r = round r of every
// clique; the differential stages of all of them behind the last round), one transfer out.  Queued, not waited for:
// clique_plans_finish waits for the copies out (an event behind them) and unpacks.  Beliefs with a handle are resident:
// taken from / delivered to their resident slot by device copies in front of / behind the rounds (a copy stage each).
struct nbp_clique_ticket {
  nbp_ctx *ctx = nullptr;
  nbp_read_token *tok = nullptr;
  std::vector<int32_t> om;
  std::vector<double *> op, ob, oi;
  std::vector<nbp_tree_belief *> dst;
  std::vector<nbp_clique_request *> reqs;  // statuses to write (batch entry)
  int32_t *status_out = nullptr;           // (single-clique entries)
  bool down = false;
  double t_launched = 0;
};
// The plan cache of a context (see clique_plans_submit): finalized programs keyed by their own descriptors with the seed fields
// blanked.  It rides with the context (nbp_ctx_attach) and is destroyed, its programs with it, at the top of nbp_ctx_destroy.
struct plan_cache {
  struct entry { uint64_t hash; std::vector<char> sig; nbp_program *prog; uint64_t used; };
  std::vector<entry> e;
  size_t cap = 64;
  uint64_t tick = 0, hits = 0, misses = 0;
  std::mutex mu;  // (clique calls on one context may come from several host threads)
  static uint64_t hash_of(const std::vector<char> &v) {
    uint64_t h = 1469598103934665603ull;  // FNV-1a over 8-byte words (the tail byte-wise)
    const size_t nw = v.size() / 8;
    const uint64_t *w = (const uint64_t *)v.data();
    for (size_t i = 0; i < nw; i++) { h ^= w[i]; h *= 1099511628211ull; }
    for (size_t i = nw * 8; i < v.size(); i++) { h ^= (unsigned char)v[i]; h *= 1099511628211ull; }
    return h;
  }
  nbp_program *find(const std::vector<char> &sig) {
    std::lock_guard<std::mutex> lk(mu);
    const uint64_t h = hash_of(sig);
    for (entry &x : e)
      if (x.hash == h && x.sig.size() == sig.size() && memcmp(x.sig.data(), sig.data(), sig.size()) == 0) {
        x.used = ++tick;
        hits++;
        return x.prog;
      }
    misses++;
    return nullptr;
  }
  bool insert(std::vector<char> &&sig, nbp_program *p) {
    std::lock_guard<std::mutex> lk(mu);
    if (cap == 0) return false;
    if (e.size() >= cap) {  // the least recently used one out: dropped once what it has queued has run
      size_t lru = 0;
      for (size_t i = 1; i < e.size(); i++) if (e[i].used < e[lru].used) lru = i;
      nbp_program_retire(e[lru].prog);
      e.erase(e.begin() + (long)lru);
    }
    const uint64_t h = hash_of(sig);
    e.push_back({h, std::move(sig), p, ++tick});
    return true;
  }
  ~plan_cache() {
    if (getenv("NBP_PLAN_CACHE_STATS")) fprintf(stderr, "[libnbp] plan cache: %llu hits, %llu misses, %zu programs kept\n", (unsigned long long)hits, (unsigned long long)misses, e.size());
    for (entry &x : e) nbp_program_destroy(x.prog);
  }
};
// What rides with a context on the host side (nbp_ctx_attach; destroyed at the top of nbp_ctx_destroy): its plan cache and the
// queue in which single-clique calls from several host threads are merged (clique_solve below).
struct combine_req {
  nbp_clique_request r;
  int32_t *status_out = nullptr;
  nbp_status rc = NBP_OK;
  std::string err;
  bool taken = false, done = false;  // taken: in a batch that is running
};
struct combiner {
  std::mutex mu;
  std::condition_variable cv, cv_lead;  // cv: "a batch is done / a lane is free"; cv_lead: "another request has queued up" (for a gathering leader only)
  std::deque<combine_req *> q;
  // lanes: lane 0 is the caller's context, the others are contexts of the library's own (same device, N, slots), created when
  // a second leader first needs one; one batch per lane at a time, batches of different lanes side by side on the device
  std::vector<nbp_ctx *> lane;
  std::vector<char> busy;
  int nbusy = 0;
  bool gathering = false;
  size_t inflight = 0;  // requests in the batches now running
  size_t expect = 1;    // callers seen when the last batch ended (running + queued)
  uint64_t batches = 0, merged = 0, widest = 0, lanes_used = 1;
};
struct ctx_host_state {
  plan_cache *pc = nullptr;
  combiner cb;
  ~ctx_host_state() {
    if (getenv("NBP_PLAN_CACHE_STATS") && cb.widest > 1)
      fprintf(stderr, "[libnbp] single-clique calls merged: %llu calls in %llu batches (widest %llu) on %llu lane(s)\n", (unsigned long long)cb.merged,
              (unsigned long long)cb.batches, (unsigned long long)cb.widest, (unsigned long long)cb.lanes_used);
    for (size_t i = 1; i < cb.lane.size(); i++)
      if (cb.lane[i]) nbp_ctx_destroy(cb.lane[i]);
    delete pc;
  }
};
static void ctx_host_state_destroy(void *o) { delete (ctx_host_state *)o; }
static ctx_host_state *host_state_of(nbp_ctx *ctx) {
  static std::mutex make_mu;
  std::lock_guard<std::mutex> lk(make_mu);
  ctx_host_state *hs = (ctx_host_state *)nbp_ctx_attached(ctx);
  if (!hs) {
    hs = new ctx_host_state();
    static const int on = getenv("NBP_PLAN_CACHE") ? atoi(getenv("NBP_PLAN_CACHE")) : 1;
    if (on) {
      hs->pc = new plan_cache();
      if (getenv("NBP_PLAN_CACHE_ENTRIES")) hs->pc->cap = (size_t)std::max(0, atoi(getenv("NBP_PLAN_CACHE_ENTRIES")));
    }
    if (nbp_ctx_attach(ctx, hs, ctx_host_state_destroy)) { delete hs; return nullptr; }
  }
  return hs;
}
static plan_cache *plan_cache_of(nbp_ctx *ctx) {
  ctx_host_state *hs = host_state_of(ctx);
  return hs ? hs->pc : nullptr;
}
static inline int resident_slot(nbp_ctx *ctx, int handle) { return nbp_ctx_slots(ctx) - handle; }
static thread_local bool t_merged_calls = false;  // this thread is running a batch merged from concurrent single-clique calls (clique_solve)
static nbp_status clique_plans_submit(nbp_ctx *ctx, std::vector<clique_plan> &plans, nbp_clique_ticket *T, bool async) {
  const int nres = nbp_ctx_resident(ctx), cap = nbp_ctx_slots(ctx) - nres;
  {
    int need = 0;
    for (const clique_plan &P : plans) need += P.nslots;
    if (need > cap)
      return hfail(NBP_ERR_RANGE, ("clique: the context has " + std::to_string(cap) + " belief slots" + (nres ? " below its resident ones" : "") + ", the call needs " +
                                   std::to_string(need) + " (nbp_clique_slots gives the bound per clique)").c_str());
  }
  {
    int offA = 0, offB = 0;
    for (const clique_plan &P : plans) offB += P.base;
    for (clique_plan &P : plans) {
      const slot_map m{P.base, offA, offB};
      for (auto &r : P.rounds) {
        for (nbp_proposal_desc &d : r.first) relocate(d, m);
        for (nbp_product_desc &d : r.second) relocate(d, m);
      }
      for (nbp_proposal_desc &d : P.deconv) relocate(d, m);
      for (clique_io &e : P.in) e.slot = m(e.slot);
      for (clique_io &e : P.out) e.slot = m(e.slot);
      offA += P.base;
      offB += P.nslots - P.base;
    }
  }
  const double t0 = seam_now();
  std::vector<int32_t> bs, bm, bn;
  std::vector<const double *> bp, bb, bi;
  std::vector<nbp_copy_desc> cin, cout;  // resident -> the call's slot, the call's slot -> resident
  for (const clique_plan &P : plans)
    for (const clique_io &e : P.in) {
      if (e.src->handle > 0) {
        if (e.src->handle > nres) return hfail(NBP_ERR_RANGE, "clique: belief handle beyond the context's resident slots (nbp_ctx_reserve_resident)");
        cin.push_back({resident_slot(ctx, e.src->handle), e.slot});
        continue;
      }
      bs.push_back(e.slot); bm.push_back(e.mani); bn.push_back(e.src->n_pts); bp.push_back(e.src->pts); bb.push_back(e.src->bw);
      bi.push_back(e.with_ipc ? e.src->ipc : nullptr);
    }
  nbp_status rc = async ? nbp_belief_write_batch_async(ctx, (int32_t)bs.size(), bs.data(), bm.data(), bp.data(), bn.data(), bb.data(), bi.data())
                        : nbp_belief_write_batch(ctx, (int32_t)bs.size(), bs.data(), bm.data(), bp.data(), bn.data(), bb.data(), bi.data());
  if (rc) return rc;
  const double t1 = seam_now();
  // ---- the stages of the batch's program, in order: [copies in] (proposals, products) x rounds [deconv] [copies out] ----------
  struct stage_buf { int32_t kind; int32_t n; std::vector<char> bytes; };
  std::vector<stage_buf> stg;
  auto add = [&](int32_t kind, const void *d, size_t n, size_t esz) {
    stg.push_back({kind, (int32_t)n, std::vector<char>((const char *)d, (const char *)d + n * esz)});
  };
  if (!cin.empty()) add(NBP_STAGE_COPIES, cin.data(), cin.size(), sizeof(nbp_copy_desc));
  size_t nr = 0;
  for (const clique_plan &P : plans) nr = std::max(nr, P.rounds.size());
  std::vector<nbp_proposal_desc> props;
  std::vector<nbp_product_desc> prods;
  for (size_t r = 0; r < nr; r++) {
    props.clear(); prods.clear();
    for (const clique_plan &P : plans)
      if (r < P.rounds.size()) {
        props.insert(props.end(), P.rounds[r].first.begin(), P.rounds[r].first.end());
        prods.insert(prods.end(), P.rounds[r].second.begin(), P.rounds[r].second.end());
      }
    add(NBP_STAGE_PROPOSALS, props.data(), props.size(), sizeof(nbp_proposal_desc));
    add(NBP_STAGE_PRODUCTS, prods.data(), prods.size(), sizeof(nbp_product_desc));
  }
  props.clear();
  for (const clique_plan &P : plans) props.insert(props.end(), P.deconv.begin(), P.deconv.end());
  if (!props.empty()) add(NBP_STAGE_DECONV, props.data(), props.size(), sizeof(nbp_proposal_desc));
  // beliefs out: to their resident slots (a copy stage: it carries the fitted bandwidth along), to the host (below)
  std::vector<int32_t> os;
  for (const clique_plan &P : plans)
    for (const clique_io &e : P.out) {
      if (e.dst->handle > 0) {
        if (e.dst->handle > nres) return hfail(NBP_ERR_RANGE, "clique: belief handle beyond the context's resident slots (nbp_ctx_reserve_resident)");
        cout.push_back({e.slot, resident_slot(ctx, e.dst->handle)});
      }
      if (e.dst->pts || e.dst->handle <= 0) {
        if (!e.dst->pts) return hfail(NBP_ERR_ARG, "clique: a delivered belief needs host buffers or a handle");
        os.push_back(e.slot); T->om.push_back(e.mani); T->op.push_back(e.dst->pts); T->ob.push_back(e.dst->bw); T->oi.push_back(e.with_ipc ? e.dst->ipc : nullptr);
        T->dst.push_back(e.dst);
      }
    }
  if (!cout.empty()) add(NBP_STAGE_COPIES, cout.data(), cout.size(), sizeof(nbp_copy_desc));
  // ---- PLAN CACHE (round 6): a batch whose program is, descriptor for descriptor, one this context has built before -- the
  // requests of a tree level that has not changed since the last walk -- runs that program again with the new seeds
  // (nbp_program_set_seeds) instead of assembling, finalizing and enqueueing ~60 launches: from its third run on the program
  // is one hipGraph launch.  The key is the PROGRAM, not the request: the fresh plan is built either way (2-3 ms of a walk,
  // on the planning pool) and compared byte for byte with its seed fields blanked, so nothing the planner looks at can be
  // missed by the key.  NBP_PLAN_CACHE=0 switches it off, NBP_PLAN_CACHE_ENTRIES (default 64) sizes it (least recently used out).
  std::vector<uint64_t> seeds;     // the seeds of the fresh plan in the order of the program's seed table
  std::vector<char> sig;           // the stages with their seed fields blanked
  plan_cache *PC = plan_cache_of(ctx);
  if (PC) {
    size_t tot = 0;
    for (const stage_buf &b : stg) tot += 8 + b.bytes.size();
    sig.reserve(tot);
    for (stage_buf &b : stg) {
      const int32_t hdr[2] = {b.kind, b.n};
      sig.insert(sig.end(), (const char *)hdr, (const char *)hdr + 8);
      const size_t at = sig.size();
      sig.insert(sig.end(), b.bytes.begin(), b.bytes.end());
      if (b.kind == NBP_STAGE_PROPOSALS || b.kind == NBP_STAGE_DECONV)
        for (int i = 0; i < b.n; i++) {
          nbp_proposal_desc *d = (nbp_proposal_desc *)(sig.data() + at) + i;
          seeds.push_back(d->seed);
          d->seed = 0;
          if (d->meas_seed) { seeds.push_back(d->meas_seed); d->meas_seed = 1; }  // (that it names a stored measurement is structure)
        }
      else if (b.kind == NBP_STAGE_PRODUCTS)
        for (int i = 0; i < b.n; i++) {
          nbp_product_desc *d = (nbp_product_desc *)(sig.data() + at) + i;
          seeds.push_back(d->seed);
          d->seed = 0;
        }
    }
  }
  nbp_program *p = PC ? PC->find(sig) : nullptr;
  const bool hit = p != nullptr;
  // (a program of its own is short-lived: retired behind its last launch when the call does not wait, destroyed otherwise; a
  //  cached one belongs to the cache)
  struct prog_guard { nbp_program *p; bool async, owned; ~prog_guard() { if (!owned || !p) return; if (async) nbp_program_retire(p); else nbp_program_destroy(p); } } guard{nullptr, async, false};
  if (hit) {
    int32_t ns = 0;
    rc = nbp_program_num_seeds(p, &ns);
    if (!rc && ns != (int32_t)seeds.size()) rc = hfail(NBP_ERR_ARG, "plan cache: the cached program's seed count is not the plan's");
    if (!rc) rc = nbp_program_set_seeds(p, seeds.data(), ns);
    if (!rc && t_merged_calls) rc = nbp_program_set_option(p, NBP_OPT_GRAPH_REPLAY, 0);
    if (rc) return rc;
  } else {
    rc = nbp_program_create(ctx, &p);
    if (rc) return rc;
    guard.p = p;
    guard.owned = true;
    rc = nbp_program_set_option(p, NBP_OPT_LAZY_BANDWIDTH, 1);
    if (!rc && async) rc = nbp_program_set_option(p, NBP_OPT_ASYNC_UPLOAD, 1);
    // (the program of a single clique or of a handful is a few launches: replayed as it is, no hipGraph -- a capture per
    //  clique program costs more than it saves, and sixteen callers' contexts would be capturing side by side)
    static const size_t graph_min = getenv("NBP_PLAN_CACHE_GRAPH_MIN") ? (size_t)atoi(getenv("NBP_PLAN_CACHE_GRAPH_MIN")) : 8;
    // (nor the batches merged from concurrent callers: their composition changes from round to round, and a capture on one lane
    //  while another lane's leader synchronises its stream is refused by the runtime -- "operation not permitted when stream is
    //  capturing", one walk in a few, whatever the capture mode)
    if (!rc && (plans.size() < graph_min || t_merged_calls)) rc = nbp_program_set_option(p, NBP_OPT_GRAPH_REPLAY, 0);
    for (const stage_buf &b : stg)
      if (!rc) rc = nbp_program_add_stage(p, b.kind, b.bytes.data(), b.n);
    if (!rc) rc = nbp_program_finalize(p);
    if (rc) return rc;
    if (PC && PC->insert(std::move(sig), p)) guard.owned = false;  // the cache keeps it
  }
  const double t2 = seam_now();
  if (!rc) rc = nbp_program_run(p, 0, -1);
  if (!rc && g_seam_sync) rc = nbp_synchronize(ctx);  // (timing mode: the wait is charged to the launches, not to the read)
  if (rc) return rc;
  const double t3 = seam_now();
  T->ctx = ctx;
  rc = nbp_belief_read_batch_begin(ctx, (int32_t)os.size(), os.data(), &T->tok);
  if (rc) return rc;
  T->t_launched = t3;
  seam_add(1, t1 - t0); seam_add(2, t2 - t1); seam_add(3, t3 - t2);
  return NBP_OK;
}
// waits for the batch of `T` (its copies out; a batch without host deliveries: the event behind its last launch), unpacks
static nbp_status clique_plans_finish(nbp_clique_ticket *T) {
  const double t3 = seam_now();
  std::vector<int32_t> on(T->dst.size());
  nbp_status rc = nbp_belief_read_batch_end(T->tok, T->om.data(), T->op.data(), on.data(), T->ob.data(), T->oi.data());
  T->tok = nullptr;
  if (rc) return rc;
  for (size_t i = 0; i < T->dst.size(); i++) T->dst[i]->n_pts = on[i];
  seam_add(4, seam_now() - t3);
  return NBP_OK;
}
static nbp_status clique_plans_run(nbp_ctx *ctx, std::vector<clique_plan> &plans) {
  nbp_clique_ticket T;
  nbp_status rc = clique_plans_submit(ctx, plans, &T, false);
  if (rc) {
    if (T.tok) nbp_belief_read_batch_end(T.tok, nullptr, nullptr, nullptr, nullptr, nullptr);
    return rc;
  }
  return clique_plans_finish(&T);
}

static nbp_status clique_particles_ok(nbp_ctx *ctx, const nbp_solver_params *sp) {
  if (sp && sp->N != nbp_ctx_particles(ctx))
    return hfail(NBP_ERR_ARG, ("clique: params.N = " + std::to_string(sp->N) + ", the context was created for N = " +
                               std::to_string(nbp_ctx_particles(ctx))).c_str());
  return NBP_OK;
}
// A small persistent pool for the planning of a batch (round 6; through round 5 every call with >= 64 requests spawned and
// joined up to seven std::threads: ~40 thread creations per walk of a 1000-variable tree).  Workers are started on first use and
// sleep on a condition variable between batches; a batch is a function of (part, parts) run once per part, part 0 on the calling
// thread.  One batch at a time (callers from several host threads queue up on the mutex: planning is short).
namespace {
struct plan_pool {
  std::mutex run_mu;                 // one batch at a time
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::vector<std::thread> workers;
  std::function<void(int, int)> job;
  int parts = 0, next = 0, pending = 0;
  uint64_t gen = 0;
  bool stop = false;
  void worker() {
    uint64_t seen = 0;
    for (;;) {
      int part, T;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_work.wait(lk, [&] { return stop || (gen != seen && next < parts); });
        if (stop) return;
        part = next++;
        T = parts;
        if (next >= parts) seen = gen;
      }
      job(part, T);
      {
        std::lock_guard<std::mutex> lk(mu);
        if (--pending == 0) cv_done.notify_all();
      }
    }
  }
  void run(int T, const std::function<void(int, int)> &f) {
    if (T <= 1) { f(0, 1); return; }
    std::lock_guard<std::mutex> batch(run_mu);
    {
      std::lock_guard<std::mutex> lk(mu);
      while ((int)workers.size() < T - 1) workers.emplace_back([this] { worker(); });
      job = f;
      parts = T;
      next = 1;          // part 0 runs here
      pending = T - 1;
      gen++;
    }
    cv_work.notify_all();
    f(0, T);
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return pending == 0; });
  }
  ~plan_pool() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv_work.notify_all();
    for (std::thread &t : workers) t.join();
  }
};
plan_pool g_plan_pool;
}  // namespace

static nbp_status clique_batch_plans(nbp_ctx *ctx, nbp_clique_request *req, int32_t n, std::vector<clique_plan> &plans) {
  const double t0 = seam_now();
  for (int i = 0; i < n; i++) {
    if (!req[i].params || !req[i].clique) return hfail(NBP_ERR_ARG, "clique batch: null params / clique");
    if (nbp_status rn = clique_particles_ok(ctx, req[i].params)) return rn;
  }
  {
    // the plans of a level are independent of each other: a few threads when there are many (a chain's leaf level: ~500).
    // The error text is thread-local, so a request that fails is planned once more on this thread to report it.
    auto build = [&](int i) { return clique_plan_build(req[i].params, req[i].clique, req[i].seed, req[i].beliefs, req[i].down != 0, req[i].diff_out, plans[(size_t)i]); };
    const unsigned hw = std::thread::hardware_concurrency();
    int nt = n / 32;
    if (nt > 8) nt = 8;
    if (hw && nt > (int)hw) nt = (int)hw;
    std::vector<nbp_status> rcs((size_t)n, NBP_OK);
    auto part = [&](int t, int T) {
      for (int i = (int)((int64_t)n * t / T); i < (int)((int64_t)n * (t + 1) / T); i++) rcs[(size_t)i] = build(i);
    };
    g_plan_pool.run(nt < 1 ? 1 : nt, part);  // (a persistent pool: no thread is created per call)
    for (int i = 0; i < n; i++)
      if (rcs[(size_t)i]) {
        plans[(size_t)i] = clique_plan();
        const nbp_status rc = build(i);
        return rc ? rc : rcs[(size_t)i];
      }
  }
  seam_add(0, seam_now() - t0); seam_add(5, 1);
  return NBP_OK;
}

// ---- single-clique calls from several host threads on ONE context: merged ------------------------------------------------
// The reference runs the cliques of a tree level as concurrent tasks, each calling its own solve (CliqueStateMachine.jl,
// SolverAPI.jl:59-97).  Through round 5 a host that kept that shape needed one context per task, and 16 such callers walked
// the 1000-variable chain in 171-198 ms -- three times the lone caller's time per call: every call pays its own transfers,
// its own five or six latency-bound launches and its own wait, and sixteen streams of those do not share the device the way
// one launch serving sixteen cliques does.  Here calls that arrive while a batch is on the device are MERGED: the first caller
// to find nobody leading leads -- it takes what has queued up (its own request among it), runs it as one batch
// (nbp_clique_solve_batch's path), marks those requests done and wakes their callers; a caller that wakes with its request
// still queued leads the next round.  Nothing in a clique's result depends on what it was batched with (streams keyed by
// (seed, pass, clique, step, factor); nbp_host.h says where a batch's bytes can differ from a single call's: nowhere on a
// chain).  A lone caller's call is a batch of one behind an uncontended mutex, as before; and concurrent single-clique calls
// on one context, which used to trample each other's slots, are now simply correct.
static void combine_run(nbp_ctx *ctx, std::vector<combine_req *> &batch, std::vector<combine_req *> &back) {
  const int n = (int)batch.size();
  std::vector<nbp_clique_request> req((size_t)n);
  for (int i = 0; i < n; i++) req[(size_t)i] = batch[(size_t)i]->r;
  std::vector<clique_plan> plans((size_t)n);
  nbp_status rc = clique_batch_plans(ctx, req.data(), n, plans);
  int take = n;
  if (!rc && n > 1) {  // as many as the context has slots for (at least one: a clique too large for it reports that itself)
    const int cap = nbp_ctx_slots(ctx) - nbp_ctx_resident(ctx);
    int need = 0;
    take = 0;
    while (take < n && (take == 0 || need + plans[(size_t)take].nslots <= cap)) need += plans[(size_t)take++].nslots;
    if (take < n) {
      back.assign(batch.begin() + take, batch.end());
      batch.resize((size_t)take);
      plans.resize((size_t)take);
    }
  }
  t_merged_calls = true;
  // (the synchronous path: the queued one -- pinned staging, the wait on the batch's own event -- measured 5-15 % slower here,
  //  two lanes' worth of short batches gain nothing from a copy the stream makes)
  if (!rc) rc = clique_plans_run(ctx, plans);
  t_merged_calls = false;
  if (rc && take > 1) {  // whose fault: each request on its own, so that every caller gets its own status and message
    for (combine_req *b : batch) {
      std::vector<clique_plan> one(1);
      b->rc = clique_batch_plans(ctx, &b->r, 1, one);
      if (!b->rc) b->rc = clique_plans_run(ctx, one);
      if (b->rc) b->err = nbp_last_error();
    }
    return;
  }
  for (combine_req *b : batch) {
    b->rc = rc;
    if (rc) b->err = nbp_last_error();
  }
}
static bool request_is_resident(const nbp_clique_request &r) {  // a belief named by handle lives in the caller's context
  for (int i = 0; i < r.clique->nvars; i++) if (r.beliefs && r.beliefs[i].handle > 0) return true;
  for (int i = 0; i < r.clique->n_diff; i++) if (r.diff_out && r.diff_out[i].handle > 0) return true;
  return false;
}
static nbp_status clique_solve(nbp_ctx *ctx, const nbp_solver_params *sp, const nbp_clique_desc *q, uint64_t seed,
                               nbp_tree_belief *bel, int32_t *status_out, bool down, nbp_tree_belief *diff_out = nullptr) {
  if (!ctx || !sp || !q) return hfail(NBP_ERR_ARG, "null argument");
  ctx_host_state *hs = host_state_of(ctx);
  if (!hs) return hfail(NBP_ERR_ARG, "clique: the context carries a foreign attachment");
  combiner &C = hs->cb;
  static const int widest = getenv("NBP_COMBINE_MAX") ? std::max(1, atoi(getenv("NBP_COMBINE_MAX"))) : 256;
  static const int nlanes = getenv("NBP_COMBINE_LANES") ? std::min(16, std::max(1, atoi(getenv("NBP_COMBINE_LANES")))) : 2;
  // the callers a batch has just released need a moment to come back with their next clique (wake up, assemble the sub
  // graph): a leader that saw more callers when the last batch ended than are queued now gives them that moment, once
  static const int gather_us = getenv("NBP_COMBINE_GATHER_US") ? std::max(0, atoi(getenv("NBP_COMBINE_GATHER_US"))) : 150;
  combine_req me;
  me.r = nbp_clique_request{sp, q, seed, bel, diff_out, down ? 1 : 0, 0};
  const bool home = request_is_resident(me.r);
  std::unique_lock<std::mutex> lk(C.mu);
  if (C.lane.empty()) { C.lane.assign((size_t)nlanes, nullptr); C.lane[0] = ctx; C.busy.assign((size_t)nlanes, 0); }
  C.q.push_back(&me);
  if (C.gathering) C.cv_lead.notify_one();
  while (!me.done) {
    // lead on a free lane (a request with resident beliefs: on the caller's own context only); one leader gathers at a time
    int L = -1;
    if (!C.gathering && !me.taken)
      for (int i = 0; i < (home ? 1 : nlanes) && L < 0; i++) if (!C.busy[(size_t)i]) L = i;
    if (L < 0) { C.cv.wait(lk); continue; }
    if (!C.lane[(size_t)L]) {  // (created under the lock: once per lane and context)
      nbp_ctx *lc = nullptr;
      const int slots = std::min(nbp_ctx_slots(ctx) - nbp_ctx_resident(ctx), 8192);
      if (nbp_ctx_create(nbp_ctx_device(ctx), nbp_ctx_particles(ctx), slots < 64 ? 64 : slots, nullptr, 0, 0, &lc)) {
        C.busy[(size_t)L] = 2;  // no memory for another lane: it stays closed, the caller takes an open one
        continue;
      }
      C.lane[(size_t)L] = lc;
      if ((uint64_t)L + 1 > C.lanes_used) C.lanes_used = (uint64_t)L + 1;
    }
    C.busy[(size_t)L] = 1;
    C.nbusy++;
    // this lane's share of the callers around: with every lane busy in turn, a batch is one lane's worth of them
    const size_t share = (C.expect + (size_t)nlanes - 1) / (size_t)nlanes;
    if (gather_us && C.q.size() < share && C.expect > 1) {
      C.gathering = true;
      const size_t want = std::min(share, (size_t)widest);
      C.cv_lead.wait_for(lk, std::chrono::microseconds(gather_us), [&] { return C.q.size() >= want; });
      C.gathering = false;
    }
    std::vector<combine_req *> batch, back;
    for (auto it = C.q.begin(); it != C.q.end() && (int)batch.size() < widest;) {
      if (L > 0 && request_is_resident((*it)->r)) { ++it; continue; }
      (*it)->taken = true;
      batch.push_back(*it);
      it = C.q.erase(it);
    }
    C.inflight += batch.size();
    C.cv.notify_all();  // (the gathering is over: another caller may lead on another lane)
    lk.unlock();
    if (!batch.empty()) combine_run(C.lane[(size_t)L], batch, back);
    lk.lock();
    for (size_t i = back.size(); i > 0; i--) { back[i - 1]->taken = false; C.q.push_front(back[i - 1]); }  // (no room in the context this time: first in line next time)
    C.inflight -= batch.size() + back.size();
    for (combine_req *b : batch) b->done = true;
    C.expect = batch.size() + C.inflight + C.q.size();
    C.batches++;
    C.merged += batch.size();
    if (batch.size() > C.widest) C.widest = batch.size();
    C.busy[(size_t)L] = 0;
    C.nbusy--;
    C.cv.notify_all();
  }
  lk.unlock();
  if (me.rc) return hfail(me.rc, me.err.c_str());
  if (status_out) *status_out = down ? NBP_CLIQ_DOWNSOLVED : NBP_CLIQ_UPSOLVED;
  return NBP_OK;
}

nbp_status nbp_clique_solve_batch(nbp_ctx *ctx, nbp_clique_request *req, int32_t n) {
  if (!ctx || (n > 0 && !req)) return hfail(NBP_ERR_ARG, "null argument");
  if (n <= 0) return NBP_OK;
  std::vector<clique_plan> plans((size_t)n);
  nbp_status rc = clique_batch_plans(ctx, req, n, plans);
  if (!rc) rc = clique_plans_run(ctx, plans);
  if (rc) return rc;
  for (int i = 0; i < n; i++) req[i].status = req[i].down ? NBP_CLIQ_DOWNSOLVED : NBP_CLIQ_UPSOLVED;
  return NBP_OK;
}

nbp_status nbp_clique_submit_batch(nbp_ctx *ctx, nbp_clique_request *req, int32_t n, nbp_clique_ticket **out) {
  if (!ctx || !out || (n > 0 && !req)) return hfail(NBP_ERR_ARG, "null argument");
  *out = nullptr;
  nbp_clique_ticket *T = new nbp_clique_ticket();
  T->ctx = ctx;
  if (n > 0) {
    std::vector<clique_plan> plans((size_t)n);
    nbp_status rc = clique_batch_plans(ctx, req, n, plans);
    if (!rc) rc = clique_plans_submit(ctx, plans, T, true);
    if (rc) {
      if (T->tok) nbp_belief_read_batch_end(T->tok, nullptr, nullptr, nullptr, nullptr, nullptr);
      delete T;
      return rc;
    }
    for (int i = 0; i < n; i++) T->reqs.push_back(&req[i]);
  }
  *out = T;
  return NBP_OK;
}
nbp_status nbp_clique_wait(nbp_clique_ticket *T) {
  if (!T) return hfail(NBP_ERR_ARG, "null argument");
  nbp_status rc = NBP_OK;
  if (T->tok) rc = clique_plans_finish(T);
  if (!rc)
    for (nbp_clique_request *r : T->reqs) r->status = r->down ? NBP_CLIQ_DOWNSOLVED : NBP_CLIQ_UPSOLVED;
  delete T;
  return rc;
}

// ---- resident beliefs from the host side ------------------------------------------------------------------------------
static nbp_status resident_slots(nbp_ctx *ctx, int32_t n, const int32_t *h, std::vector<int32_t> &slots) {
  if (!ctx || (n > 0 && !h)) return hfail(NBP_ERR_ARG, "null argument");
  const int nres = nbp_ctx_resident(ctx);
  slots.resize((size_t)(n > 0 ? n : 0));
  for (int i = 0; i < n; i++) {
    if (h[i] < 1 || h[i] > nres) return hfail(NBP_ERR_RANGE, "resident belief: handle outside 1 .. nbp_ctx_resident");
    slots[(size_t)i] = resident_slot(ctx, h[i]);
  }
  return NBP_OK;
}
nbp_status nbp_resident_write(nbp_ctx *ctx, int32_t n, const int32_t *handles, const int32_t *manifolds, const nbp_tree_belief *b) {
  std::vector<int32_t> slots, np;
  nbp_status rc = resident_slots(ctx, n, handles, slots);
  if (rc || n <= 0) return rc;
  if (!manifolds || !b) return hfail(NBP_ERR_ARG, "null argument");
  std::vector<const double *> p, w, c;
  for (int i = 0; i < n; i++) { p.push_back(b[i].pts); w.push_back(b[i].bw); c.push_back(b[i].ipc); np.push_back(b[i].n_pts); }
  return nbp_belief_write_batch_async(ctx, n, slots.data(), manifolds, p.data(), np.data(), w.data(), c.data());
}
nbp_status nbp_resident_read(nbp_ctx *ctx, int32_t n, const int32_t *handles, const int32_t *manifolds, nbp_tree_belief *b) {
  std::vector<int32_t> slots, np((size_t)(n > 0 ? n : 0));
  nbp_status rc = resident_slots(ctx, n, handles, slots);
  if (rc || n <= 0) return rc;
  if (!manifolds || !b) return hfail(NBP_ERR_ARG, "null argument");
  std::vector<double *> p, w, c;
  for (int i = 0; i < n; i++) { p.push_back(b[i].pts); w.push_back(b[i].bw); c.push_back(b[i].ipc); }
  rc = nbp_belief_read_batch(ctx, n, slots.data(), manifolds, p.data(), np.data(), w.data(), c.data());
  if (!rc) for (int i = 0; i < n; i++) b[i].n_pts = np[(size_t)i];
  return rc;
}
nbp_status nbp_resident_copy(nbp_ctx *ctx, int32_t n, const int32_t *src, const int32_t *dst, int32_t points_only) {
  std::vector<int32_t> a, b;
  nbp_status rc = resident_slots(ctx, n, src, a);
  if (!rc) rc = resident_slots(ctx, n, dst, b);
  if (rc || n <= 0) return rc;
  std::vector<nbp_copy_desc> cd((size_t)n);
  for (int i = 0; i < n; i++) cd[(size_t)i] = {a[(size_t)i], b[(size_t)i]};
  return nbp_run_copies_async(ctx, cd.data(), n, points_only);
}

nbp_status nbp_clique_upsolve(nbp_ctx *ctx, const nbp_solver_params *sp, const nbp_clique_desc *q, uint64_t seed,
                              nbp_tree_belief *bel, int32_t *status_out) {
  return clique_solve(ctx, sp, q, seed, bel, status_out, false);
}
nbp_status nbp_clique_upsolve_joint(nbp_ctx *ctx, const nbp_solver_params *sp, const nbp_clique_desc *q, uint64_t seed,
                                    nbp_tree_belief *bel, nbp_tree_belief *diff_out, int32_t *status_out) {
  return clique_solve(ctx, sp, q, seed, bel, status_out, false, diff_out);
}
nbp_status nbp_clique_downsolve(nbp_ctx *ctx, const nbp_solver_params *sp, const nbp_clique_desc *q, uint64_t seed,
                                nbp_tree_belief *bel, int32_t *status_out) {
  return clique_solve(ctx, sp, q, seed, bel, status_out, true);
}

nbp_status nbp_clique_seam_times(double *out, int32_t mode) {
  if (out) for (int i = 0; i < 6; i++) out[i] = g_seam_s[i].load(std::memory_order_relaxed);
  if (mode >= 1) for (int i = 0; i < 6; i++) g_seam_s[i].store(0.0, std::memory_order_relaxed);
  if (mode >= 1) g_seam_sync.store(mode == 2);
  return NBP_OK;
}

int32_t nbp_tree_num_stages(const nbp_tree *t) { return t ? (int32_t)t->stages.size() : 0; }
nbp_status nbp_tree_stage(const nbp_tree *t, int32_t s, int32_t *kind, int32_t *n, void *out, int64_t cap) {
  if (!t || s < 0 || s >= (int)t->stages.size()) return hfail(NBP_ERR_RANGE, "stage index");
  const Stage &st = t->stages[s];
  if (kind) *kind = st.kind;
  if (n) *n = st.n;
  if (out && cap > 0) memcpy(out, st.bytes.data(), std::min<size_t>((size_t)cap, st.bytes.size()));
  return NBP_OK;
}

}  // extern "C"
