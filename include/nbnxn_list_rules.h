// Rules of the pair-list builders that the host builder (nbnxn_host.cpp) and the device search (nbnxn_search.hip)
// must apply identically, so that the two produce the same list entry for entry.
#ifndef NBNXN_LIST_RULES_H
#define NBNXN_LIST_RULES_H

#ifdef __HIPCC__
#define NBX_RULE_FN static inline __host__ __device__
#else
#define NBX_RULE_FN static inline
#endif

// Second key of sort_sci.  The reference orders the entries of a GPU list by their cj4 count alone, descending and
// stable (nbnxn_search.c:5517-5580); entries of one count stay in the order the builder made them, whatever work they
// hold.  The GPU deals entries to wave slots in list order, and once the list has run out every SIMD finishes what it
// holds at the pace of a lone wave: the launch ends one LAST entry after that, and a full cj4 group (32 cluster pairs)
// takes three times as long as a sparse one.  So among entries of one group count those with more cluster pairs (set bits
// of the interaction masks as built) go first.  Only for entries of up to two thirds of the clip m (= twice the mean
// group count): the large entries all start together in the first round of waves, where their order decides nothing
// but who shares a SIMD with whom, and sorting them made the reaction-field kernels slower.  Measured at 97 k atoms
// (tools/tail_sort_probe.py, profiles/r3_k_tail_sort.txt): Ewald V+F -3.2 %, Ewald F -3 %, RF V+F -1 %, RF F -0.1 %.
// The key is the pair count in NBX_SCI_PAIR_KEYS buckets (a counting sort on the device), 0 = most pairs.
#define NBX_SCI_PAIR_KEYS 64
NBX_RULE_FN int nbx_sci_pair_key(int ngroups, int npair, int m)
{
    const int lim = 2 * m / 3;
    if (ngroups > lim) { return NBX_SCI_PAIR_KEYS - 1; }
    const int div = lim + 1 > 1 ? (lim + 1) / 2 : 1;
    const int q   = npair / div;
    return NBX_SCI_PAIR_KEYS - 1 - (q < NBX_SCI_PAIR_KEYS - 1 ? q : NBX_SCI_PAIR_KEYS - 1);
}

#endif
