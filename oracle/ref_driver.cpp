// ref_driver.cpp -- thin C entry points over the REAL reference translation units.
//
// TEST INFRASTRUCTURE ONLY.  This file is this repo's own code; it is compiled together with
// the reference's own sources where they lie (/root/reference/src/{MurmurHash3,gfa-io,gfa-base,
// kalloc,misc,options,sys}.cpp -- the seven TUs that build without Gurobi) into
// oracle/_ref/libphi_ref.so by oracle/Makefile.  It is used to pin the CPU restatement
// (oracle/phi_oracle.c, oracle/oracle.py) and the product's GFA reader against the reference.
// ILP_index.cpp and main.cpp need gurobi_c++.h, which this image lacks: they are unbuildable
// here and are not part of this library.
#include <stdint.h>
#include <string.h>
#include <string>
#include "gfa.h"            // /root/reference/src/gfa.h
#include "MurmurHash3.h"    // /root/reference/src/MurmurHash3.h
#include "PHIpriv.h"        // /root/reference/src/PHIpriv.h: declares get_hap_name (misc.cpp:58)
#include <zlib.h>
#include <vector>
#include "kseq.h"           // /root/reference/src/kseq.h, instantiated exactly as ILP_index.cpp:8 does
KSEQ_INIT(gzFile, gzread)

extern "C" {

// hash128_to_64 of ILP_index.cpp:10-18, on top of the reference's MurmurHash3_x64_128.
uint64_t ref_hash128_to_64(const void *key, int len)
{
    uint64_t out[2];
    MurmurHash3_x64_128(key, len, 0, out);
    return out[0] ^ out[1];
}

void *ref_gfa_read(const char *fn) { return gfa_read(fn); }

uint32_t ref_gfa_n_seg(void *g_) { return ((gfa_t *)g_)->n_seg; }
uint32_t ref_gfa_n_walk(void *g_) { return ((gfa_t *)g_)->n_walk; }
const char *ref_gfa_seg_name(void *g_, uint32_t s) { return ((gfa_t *)g_)->seg[s].name; }
const char *ref_gfa_seg_seq(void *g_, uint32_t s) { return ((gfa_t *)g_)->seg[s].seq; }
int32_t ref_gfa_seg_len(void *g_, uint32_t s) { return ((gfa_t *)g_)->seg[s].len; }

// arcs leaving oriented vertex v (v = seg<<1|strand), in the reference's stored order
uint32_t ref_gfa_arc_n(void *g_, uint32_t v) { gfa_t *g = (gfa_t *)g_; return gfa_arc_n(g, v); }
uint32_t ref_gfa_arc_w(void *g_, uint32_t v, uint32_t i)
{
    gfa_t *g = (gfa_t *)g_;
    return gfa_arc_a(g, v)[i].w;
}

const char *ref_gfa_walk_sample(void *g_, uint32_t w) { return ((gfa_t *)g_)->walk[w].sample; }
int32_t ref_gfa_walk_hap(void *g_, uint32_t w) { return ((gfa_t *)g_)->walk[w].hap; }
int32_t ref_gfa_walk_n_v(void *g_, uint32_t w) { return ((gfa_t *)g_)->walk[w].n_v; }
const uint32_t *ref_gfa_walk_v(void *g_, uint32_t w) { return ((gfa_t *)g_)->walk[w].v; }

// get_hap_name (misc.cpp:58-87) into a caller buffer
int ref_get_hap_name(const char *gfa_name, const char *reads_name, char *out, int cap)
{
    std::string name;
    get_hap_name((char *)gfa_name, (char *)reads_name, name);
    if ((int)name.size() + 1 > cap) return -1;
    memcpy(out, name.c_str(), name.size() + 1);
    return (int)name.size();
}


// ILP_index::read_ip_reads (ILP_index.cpp:313-328) on the reference's own kseq: every record's name and
// sequence, NUL-separated, into caller buffers; returns the number of records (-1: cannot open, -2: too small)
int64_t ref_read_reads(const char *fn, char *names, int64_t names_cap, char *seqs, int64_t seqs_cap, int64_t *seq_off, int64_t off_cap)
{
    gzFile fp = gzopen(fn, "r");
    if (!fp) return -1;
    kseq_t *seq = kseq_init(fp);
    int64_t n = 0, nn = 0, ns = 0;
    int l;
    seq_off[0] = 0;
    while ((l = kseq_read(seq)) >= 0) {
        const int64_t ln = (int64_t)seq->name.l, ls = (int64_t)seq->seq.l;
        if (nn + ln + 1 > names_cap || ns + ls > seqs_cap || n + 1 >= off_cap) { n = -2; break; }
        memcpy(names + nn, seq->name.s, (size_t)ln); names[nn + ln] = 0; nn += ln + 1;
        if (ls) memcpy(seqs + ns, seq->seq.s, (size_t)ls);
        ns += ls;
        seq_off[++n] = ns;
    }
    kseq_destroy(seq);
    gzclose(fp);
    return n;
}
}
