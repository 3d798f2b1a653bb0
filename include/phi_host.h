/*
 * phi_host.h -- C ABI of the host-side feeders of the hot path (libphi_host.so, CPU only).
 *
 * These are the steps on either side of the GPU path (SURVEY.md section 8f "next"):
 *   phi_gfa_read     gfa_read() + ILP_index::read_gfa()   src/gfa-io.cpp:462-508, src/ILP_index.cpp:20-155
 *   phi_reads_read   ILP_index::read_ip_reads()           src/ILP_index.cpp:313-328 (kseq FASTA/FASTQ, gz)
 *   phi_reads_stream_*  the same records in chunks, for files larger than one buffer; phi_reads_stream_open_blocks: over text
 *                       in memory + blocks a callback hands over (the rest of a stream whose beginning the device has taken)
 *   phi_text_stream_*   the (inflated) text of a reads file as it is, for phi_add_reads_text (phi_amd.h): the records are
 *                       found on the device, no host core parses a regular FASTA / FASTQ file
 *   phi_hap_name     get_hap_name()                       src/misc.cpp:58-87
 *   phi_write_fasta  the FASTA writer                     src/ILP_index.cpp:1590-1598
 * The arrays phi_gfa_read returns are exactly the arguments of phi_set_graph (phi_amd.h).
 * All functions return 0 or a negative code and never call exit(); *err receives a message.
 */
#ifndef PHI_HOST_H
#define PHI_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct phi_graph phi_graph;
typedef struct phi_reads phi_reads;

enum {
    PHI_HOST_OK = 0,
    PHI_HOST_ERR_IO = -1,       /* file cannot be opened / read (main.cpp:102-105 returns 1) */
    PHI_HOST_ERR_WALK = -2,     /* a walk holds a reverse-strand vertex (ILP_index.cpp:104-107 exits) */
    PHI_HOST_ERR_CYCLE = -3,    /* Kahn's algorithm did not reach every vertex: graph is not acyclic */
    PHI_HOST_ERR_INVALID = -4,
    PHI_HOST_ERR_UNSUPPORTED = -5 /* a link onto the reverse strand of its target: the reference's own adjacency for it depends on
                                     the line's place in the file (gfa-base.cpp:269-303): refused, not guessed */
};

/* Parse S/L/W lines (plain or gzip), flip walks as gfa_walk_flip does, complete arcs with their
 * complements, keep forward-strand arcs, build walks / names / Kahn topological order. */
int phi_gfa_read(const char *path, phi_graph **out, char *err, int err_cap);
/* The same with the walks left as TEXT, for a caller that resolves them on the device (include/phi_amd.h phi_walk_text_*: at
 * chromosome scale the W-lines are 96 % of the file and resolving them was the reader's largest stage).  on_text (optional) is
 * called on a thread of the reader's own as soon as the walk fields are known -- before the segment names are --, with slices
 * of the mapped file (valid until the walks are resolved either way); it is joined before the call returns.
 *   phi_graph_walks_deferred   1 while the walks are text
 *   phi_graph_walk_texts       the walk fields again (optional tags may follow a walk); returns their number
 *   phi_graph_name_index       the name table's direct index: names are <prefix><canonical decimal> -> num2id[number];
 *                              PHI_HOST_ERR_UNSUPPORTED when some name is of another form or a W-line stands before an S-line
 *                              (the device path does not take such a file: phi_graph_resolve_walks)
 *   phi_graph_resolve_walks    the host path after all, with the reference's rules (steps naming no segment left out, walks
 *                              flipped by majority strand, a reverse vertex left over is PHI_HOST_ERR_WALK)
 *   phi_graph_set_walk_off     the device resolved them: their offsets [n_walks + 1]; the text is let go
 * phi_graph_walk_vtx stays NULL for walks resolved on the device. */
typedef struct { const char *text; int64_t n; } phi_host_walk_text;
typedef void (*phi_walk_text_fn)(void *user, const phi_host_walk_text *walks, int32_t n_walks);
int phi_gfa_read_deferred(const char *path, phi_graph **out, phi_walk_text_fn on_text, void *user, char *err, int err_cap);
int phi_graph_walks_deferred(const phi_graph *g);
int phi_graph_walk_texts(const phi_graph *g, phi_host_walk_text *out, int32_t cap);
int phi_graph_name_index(const phi_graph *g, const char **prefix, int32_t *prefix_n, const int32_t **num2id, int64_t *n_num);
int phi_graph_resolve_walks(phi_graph *g, char *err, int err_cap);
int phi_graph_set_walk_off(phi_graph *g, const int64_t *walk_off);
void phi_graph_free(phi_graph *g);

int32_t phi_graph_n_vtx(const phi_graph *g);
int32_t phi_graph_n_walks(const phi_graph *g);
int64_t phi_graph_n_edges(const phi_graph *g);
const char *phi_graph_seq_concat(const phi_graph *g);      /* node_seq, original case */
const int64_t *phi_graph_seq_off(const phi_graph *g);      /* [n_vtx+1] */
const int64_t *phi_graph_adj_off(const phi_graph *g);      /* [n_vtx+1] */
const int32_t *phi_graph_adj(const phi_graph *g);
const int64_t *phi_graph_walk_off(const phi_graph *g);     /* [n_walks+1] */
const int32_t *phi_graph_walk_vtx(const phi_graph *g);
const int32_t *phi_graph_topo_rank(const phi_graph *g);    /* top_order_map */
const char *phi_graph_hap_name(const phi_graph *g, int32_t walk);   /* sample + "." + hap (:98) */
const char *phi_graph_seg_name(const phi_graph *g, int32_t vtx);

/* FASTA/FASTQ reader with kseq's record rules (multi-line FASTA, '+' quality blocks). */
int phi_reads_read(const char *path, phi_reads **out, char *err, int err_cap);
void phi_reads_free(phi_reads *r);
int64_t phi_reads_count(const phi_reads *r);
const char *phi_reads_bases(const phi_reads *r);           /* concatenated sequences */
const int64_t *phi_reads_off(const phi_reads *r);          /* [count+1] */
const char *phi_reads_name(const phi_reads *r, int64_t i);

/* The same reader, streaming: phi_reads_stream_next parses the next records into the
 * caller's buffers -- bases[bases_cap] (e.g. pinned memory the device copy reads while the next chunk
 * is parsed) and off[reads_cap + 1], off[0] = 0 -- never splitting a record, and returns their number:
 * 0 at the end of the file, negative on error (a single read longer than bases_cap). */
typedef struct phi_reads_stream phi_reads_stream;
int phi_reads_stream_open(const char *path, phi_reads_stream **out, char *err, int err_cap);
int64_t phi_reads_stream_next(phi_reads_stream *s, char *bases, int64_t bases_cap, int64_t *off, int64_t reads_cap,
                              char *err, int err_cap);
int64_t phi_reads_stream_reads(const phi_reads_stream *s);  /* records / bases returned so far */
int64_t phi_reads_stream_bases(const phi_reads_stream *s);
void phi_reads_stream_close(phi_reads_stream *s);

/* The same records from text that is already in memory followed by blocks a callback hands over: the way to finish, on
 * this exact state machine, a stream whose beginning the device has taken (phi_add_reads_text, phi_amd.h).  next returns
 * the length of the next block and its address (valid until the next call), 0 at the end, negative on error; it may be NULL. */
typedef int64_t (*phi_text_block_fn)(void *user, const char **block);
/* stream_offset = bytes of the stream before prefix[0] (what the device took: *n_taken of phi_reads_text_end): kseq reads in
 * blocks of 65 536 bytes and what it makes of a file's very last bytes depends on the file's size modulo that (kseq.h:81,113,242). */
int phi_reads_stream_open_blocks(const char *prefix, int64_t n_prefix, phi_text_block_fn next, void *user, int64_t stream_offset,
                                 phi_reads_stream **out, char *err, int err_cap);

/* The (inflated) text of a reads file as it is -- no parsing on the host: the bytes go to phi_add_reads_text, which finds the
 * records on the device.  phi_text_stream_read fills buf with the next bytes (plain files: several preads at once, so that
 * a pinned buffer fills at more than one core's copy rate) and returns their number: 0 at the end, negative on error. */
typedef struct phi_text_stream phi_text_stream;
int phi_text_stream_open(const char *path, phi_text_stream **out, char *err, int err_cap);
int64_t phi_text_stream_read(phi_text_stream *s, char *buf, int64_t cap, char *err, int err_cap);
void phi_text_stream_close(phi_text_stream *s);

/* Record id of the output: basename(gfa) minus extension + "_" + basename(reads), minus the last
 * extension of the whole string.  Returns the length or -1 if cap is too small. */
int phi_hap_name(const char *gfa_path, const char *reads_path, char *out, int cap);

/* ">" name " LN:" len, then the sequence in 80-column lines. */
int phi_write_fasta(const char *path, const char *name, const char *seq, int64_t len);

#ifdef __cplusplus
}
#endif
#endif
