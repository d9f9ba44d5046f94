// hg38 data path: the step in front of the model.
//   * cad_tokenize_mlm  (GPU): ASCII bases -> token ids, optional reverse complement, N -> [PAD], left padding, and the
//     MLM corruption, in one pass: 1 byte read and 16 bytes written per position (pure HBM streaming, integer path).
//   * cad_fasta_*       (host): memory-mapped FASTA with a .fai-style index built on open; slices are copied out without
//     their line breaks into caller-owned (pinned) buffers.
//   * cad_hg38_interval (host): the interval arithmetic of FastaInterval.__call__.
// Randomness is counter-based (Philox4x32-10 keyed by the seed; counter = position, row, offset), so a batch is
// reproducible and independent of launch geometry; oracle/data_oracle.py restates it in numpy bit for bit.
#include "cad_common.h"

#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string>
#include <vector>

namespace {

struct Philox {
    uint32_t c[4];
};
__host__ __device__ __forceinline__ void philox_round(uint32_t* c, uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1, c[3] = (uint32_t)p0, c[0] = n0, c[2] = n2;
}
__host__ __device__ __forceinline__ Philox philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                         uint32_t k1) {
    Philox r = {{c0, c1, c2, c3}};
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        philox_round(r.c, k0, k1);
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
    return r;
}

// ASCII -> id (upper-cased, as CaduceusTokenizer._tokenize does); unknown characters -> unk
__device__ __forceinline__ int base_id(uint8_t ch, const cad_mlm_args& a) {
    const uint8_t u = (ch >= 'a' && ch <= 'z') ? (uint8_t)(ch - 32) : ch;
    return u == 'A' ? a.base_ids[0] : u == 'C' ? a.base_ids[1] : u == 'G' ? a.base_ids[2] : u == 'T' ? a.base_ids[3]
         : u == 'N' ? a.n_id : a.unk_id;
}
// complement of a CHARACTER (string_reverse_complement: A<->T, C<->G, case kept, anything else unchanged)
__device__ __forceinline__ uint8_t comp_char(uint8_t ch) {
    switch (ch) {
        case 'A': return 'T';
        case 'T': return 'A';
        case 'C': return 'G';
        case 'G': return 'C';
        case 'a': return 't';
        case 't': return 'a';
        case 'c': return 'g';
        case 'g': return 'c';
        default: return ch;
    }
}

__global__ void tokenize_mlm_kernel(cad_mlm_args a) {
    const int64_t total = a.B * a.L;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int64_t b = idx / a.L, p = idx - b * a.L;
        const int64_t full = a.lengths ? a.lengths[b] : a.L;  // bases of this row
        // a row longer than L keeps its FIRST L tokens: the reference tokenizer truncates on the right
        // (truncation=True, default truncation_side; hg38_dataset.py:190-200) AFTER the reverse complement (:179-180),
        // so the RC index below runs over the full length
        const int64_t len = full < a.L ? full : a.L;           // tokens of this row, left-padded to L
        const int64_t q = p - (a.L - len);                     // position inside the (truncated) sequence
        int id = a.pad_id;
        if (q >= 0) {
            const bool rc = a.rc_flags && a.rc_flags[b];
            const uint8_t raw = a.bases[b * a.ld_bases + (rc ? full - 1 - q : q)];
            id = base_id(rc ? comp_char(raw) : raw, a);
            if (id == a.n_id) id = a.pad_id;  // replace_value(N -> pad): ignored by the loss
        }
        int in_id = id, label = a.pad_id;
        if (a.labels && q >= 0) {
            const uint32_t sid = (uint32_t)(a.row_ids ? a.row_ids[b] : b);  // random stream of this row
            const Philox r = philox4x32_10((uint32_t)p, (uint32_t)((uint64_t)p >> 32), sid, (uint32_t)a.offset,
                                           (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
            if (r.c[0] < a.thr_mask) {              // Bernoulli(mlm_probability): a target
                label = id;
                if (r.c[1] < 0xCCCCCCCCu) {         // 80 %: [MASK]
                    in_id = a.mask_id;
                } else if (r.c[2] < 0x80000000u) {  // half of the rest: a uniformly random token id
                    in_id = (int)(((uint64_t)r.c[3] * (uint64_t)a.vocab) >> 32);
                }
            }
        }
        a.input_ids[idx] = in_id;
        if (a.labels) a.labels[idx] = label;
    }
}

// ---- host: FASTA store -------------------------------------------------------------------------------------------------
struct FastaSeq {
    std::string name;
    int64_t length;      // bases
    int64_t offset;      // byte offset of the first base
    int64_t line_bases;  // bases per line
    int64_t line_width;  // bytes per line including the line break
};
struct FastaStore {
    int fd = -1;
    const char* data = nullptr;
    size_t size = 0;
    std::vector<FastaSeq> seqs;
};

bool build_index(FastaStore* st) {
    const char* d = st->data;
    const size_t n = st->size;
    size_t i = 0;
    while (i < n) {
        if (d[i] != '>') return false;
        size_t e = i + 1;
        while (e < n && d[e] != '\n') ++e;
        size_t ne = i + 1;
        while (ne < e && d[ne] != ' ' && d[ne] != '\t' && d[ne] != '\r') ++ne;
        FastaSeq s;
        s.name.assign(d + i + 1, ne - (i + 1));
        s.offset = (int64_t)(e + 1 < n ? e + 1 : n);
        s.length = 0, s.line_bases = 0, s.line_width = 0;
        size_t p = (size_t)s.offset;
        bool first = true, short_seen = false;
        while (p < n && d[p] != '>') {
            const char* nl = (const char*)memchr(d + p, '\n', n - p);
            const size_t le = nl ? (size_t)(nl - d) : n;
            size_t bases = le - p;
            if (bases > 0 && d[le - 1] == '\r') --bases;
            const size_t width = (nl ? le + 1 : le) - p;
            if (first) {
                s.line_bases = (int64_t)bases, s.line_width = (int64_t)width, first = false;
            } else if (bases > 0) {
                if (short_seen || (int64_t)bases > s.line_bases) return false;  // only the last line may be short
            }
            if ((int64_t)bases < s.line_bases) short_seen = true;
            s.length += (int64_t)bases;
            p = nl ? le + 1 : n;
        }
        if (s.line_bases == 0) s.line_bases = 1, s.line_width = 1;
        st->seqs.push_back(s);
        i = p;
    }
    return true;
}

}  // namespace

extern "C" int cad_tokenize_mlm(const cad_mlm_args* a, void* stream) {
    CAD_CHECK_ARG(a && a->bases && a->input_ids && a->B > 0 && a->L > 0 && a->ld_bases >= 0);
    CAD_CHECK_ARG(a->vocab > 0 && a->pad_id >= 0 && a->mask_id >= 0 && a->unk_id >= 0 && a->n_id >= 0);
    const int64_t total = a->B * a->L;
    int64_t nb = (total + 255) / 256;
    if (nb > 65535) nb = 65535;
    CAD_LAUNCH(tokenize_mlm_kernel, dim3((unsigned)nb), dim3(256), 0, stream, *a);
    return cad_after_launch();
}

extern "C" uint32_t cad_mlm_threshold(double probability) {
    if (!(probability > 0.0)) return 0u;
    const double t = probability * 4294967296.0;
    return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}

extern "C" int cad_hg38_interval(int64_t start, int64_t end, int64_t max_length, int64_t i_shift, int64_t chrom_len,
                                 int64_t* out_start, int64_t* out_end) {
    const int64_t MAX_ALLOWED = 1 << 20;  // hg38_dataset.py:15
    CAD_CHECK_ARG(out_start && out_end && max_length > 0);
    if (max_length > MAX_ALLOWED) return CAD_ERR_UNSUPPORTED;  // the reference raises ValueError
    if (max_length < MAX_ALLOWED) {
        CAD_CHECK_ARG(MAX_ALLOWED % max_length == 0);
        end = start + (i_shift + 1) * max_length;
        start = start + i_shift * max_length;
    }
    if (end > chrom_len) {  // shift the interval down
        start -= end - chrom_len;
        end = chrom_len;
    }
    if (start < 0) {  // shift it up
        end -= start;
        start = 0;
    }
    if (end > chrom_len) {  // chromosome shorter than the interval
        start = chrom_len - max_length;
        end = chrom_len;
    }
    if (start < 0) start = 0;
    *out_start = start, *out_end = end;
    return CAD_OK;
}

extern "C" int cad_fasta_open(const char* path, void** handle) {
    CAD_CHECK_ARG(path && handle);
    FastaStore* st = new FastaStore();
    st->fd = open(path, O_RDONLY);
    struct stat sb;
    if (st->fd < 0 || fstat(st->fd, &sb) != 0 || sb.st_size == 0) {
        if (st->fd >= 0) close(st->fd);
        delete st;
        return CAD_ERR_BAD_ARG;
    }
    st->size = (size_t)sb.st_size;
    void* m = mmap(nullptr, st->size, PROT_READ, MAP_PRIVATE, st->fd, 0);
    if (m == MAP_FAILED) {
        close(st->fd);
        delete st;
        return CAD_ERR_BAD_ARG;
    }
    st->data = (const char*)m;
    if (!build_index(st)) {
        munmap(m, st->size);
        close(st->fd);
        delete st;
        return CAD_ERR_UNSUPPORTED;  // not a FASTA file with uniform line lengths
    }
    *handle = st;
    return CAD_OK;
}

extern "C" int cad_fasta_close(void* handle) {
    FastaStore* st = (FastaStore*)handle;
    if (!st) return CAD_OK;
    if (st->data) munmap((void*)st->data, st->size);
    if (st->fd >= 0) close(st->fd);
    delete st;
    return CAD_OK;
}

extern "C" int64_t cad_fasta_num_seqs(void* handle) { return handle ? (int64_t)((FastaStore*)handle)->seqs.size() : -1; }

extern "C" const char* cad_fasta_seq_name(void* handle, int64_t i) {
    FastaStore* st = (FastaStore*)handle;
    return (st && i >= 0 && i < (int64_t)st->seqs.size()) ? st->seqs[(size_t)i].name.c_str() : nullptr;
}

extern "C" int64_t cad_fasta_seq_len(void* handle, int64_t i) {
    FastaStore* st = (FastaStore*)handle;
    return (st && i >= 0 && i < (int64_t)st->seqs.size()) ? st->seqs[(size_t)i].length : -1;
}

extern "C" int64_t cad_fasta_find(void* handle, const char* name) {
    FastaStore* st = (FastaStore*)handle;
    if (!st || !name) return -1;
    for (size_t i = 0; i < st->seqs.size(); ++i)
        if (st->seqs[i].name == name) return (int64_t)i;
    return -1;
}

extern "C" int cad_fasta_fetch(void* handle, int64_t seq, int64_t start, int64_t end, uint8_t* out) {
    FastaStore* st = (FastaStore*)handle;
    CAD_CHECK_ARG(st && out && seq >= 0 && seq < (int64_t)st->seqs.size());
    const FastaSeq& s = st->seqs[(size_t)seq];
    CAD_CHECK_ARG(start >= 0 && start <= end && end <= s.length);
    int64_t pos = start;
    while (pos < end) {
        const int64_t line = pos / s.line_bases, col = pos % s.line_bases;
        int64_t n = s.line_bases - col;
        if (n > end - pos) n = end - pos;
        memcpy(out + (pos - start), st->data + s.offset + line * s.line_width + col, (size_t)n);
        pos += n;
    }
    return CAD_OK;
}
