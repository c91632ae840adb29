// text_docs.cpp -- the reference's corpus files parsed straight into CSR (host only; see
// host_common.h).
//
// The reference's corpus format (python/utils/load_documents.py:6-69): one document per line,
// "<n> id:cnt id:cnt ..."; the loader does `for word in line.split()[1:]: wid, wct =
// word.split(':')` and int() on both.  Here a file is mapped (or a caller's buffer taken), cut
// into pieces at line ends and parsed by the host threads.  Anything but [+-]digits:[+-]digits
// tokens (or values outside int32) is reported with its line number; the Python mirror then
// re-reads the text the slow way and raises what the reference would.
#include "host_common.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <memory>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

#include "../../include/trlda_hip.h"

using trlda_host::fail;
using trlda_host::host_pool;

struct trlda_docs {
    std::vector<int64_t> offsets;     // num_docs + 1
    std::vector<int32_t> ids, cnts;
};

namespace {

struct TextPiece {
    // raw buffers sized for the worst case (a token "1:1 " is four bytes, a line one): filled
    // through pointers, no per-token capacity checks
    std::unique_ptr<int32_t[]> lens, ids, cnts;
    size_t n_docs = 0, n_nz = 0;
    int64_t bad_line = -1;            // line (within the piece) of the first malformed token
    int64_t lines = 0;
};

// one [+-]digits run; returns false if malformed or outside int32
inline bool parse_int(const char *&p, const char *end, int32_t *out)
{
    const char *q = p;
    bool neg = false;
    if (q < end && (*q == '+' || *q == '-')) {
        neg = *q == '-';
        ++q;
    }
    const char *d0 = q;
    uint64_t v = 0;
    while (q < end) {
        const unsigned c = (unsigned)(*q - '0');
        if (c > 9)
            break;
        v = v * 10 + c;
        ++q;
    }
    const long nd = q - d0;
    if (nd == 0 || nd > 10 || v > (uint64_t)INT32_MAX + (neg ? 1 : 0))
        return false;
    *out = neg ? (int32_t)(-(int64_t)v) : (int32_t)v;
    p = q;
    return true;
}

// lines of [p, end); the last one may lack its newline
void parse_piece(const char *p, const char *end, TextPiece &out)
{
    const size_t bytes = (size_t)(end - p);
    out.ids.reset(new int32_t[bytes / 4 + 2]);
    out.cnts.reset(new int32_t[bytes / 4 + 2]);
    // (one length per line: counted, not bounded by the byte count)
    size_t lines = 1;
    for (const char *c = p; (c = static_cast<const char *>(std::memchr(c, '\n', (size_t)(end - c)))); ++c)
        ++lines;
    out.lens.reset(new int32_t[lines + 1]);
    int32_t *ids = out.ids.get(), *cnts = out.cnts.get(), *lens = out.lens.get();
    size_t nz = 0, nd = 0;
    while (p < end) {
        // the first token of the line (line.split()[0]): skipped whatever it is
        while (p < end && *p != '\n' && (*p == ' ' || (*p >= '\t' && *p <= '\r')))
            ++p;
        while (p < end && !(*p == ' ' || (*p >= '\t' && *p <= '\r')))
            ++p;
        const size_t nz0 = nz;
        for (;;) {
            while (p < end && *p != '\n' && (*p == ' ' || (*p >= '\t' && *p <= '\r')))
                ++p;
            if (p >= end || *p == '\n')
                break;
            int32_t a, b;
            if (!parse_int(p, end, &a) || p >= end || *p != ':' || (++p, !parse_int(p, end, &b)) ||
                (p < end && !(*p == ' ' || (*p >= '\t' && *p <= '\r')))) {
                out.bad_line = (int64_t)nd;
                out.n_docs = nd;
                out.n_nz = nz0;
                out.lines = (int64_t)nd;
                return;
            }
            ids[nz] = a;
            cnts[nz] = b;
            ++nz;
        }
        lens[nd++] = (int32_t)(nz - nz0);
        if (p < end)
            ++p;                                     // the newline
    }
    out.n_docs = nd;
    out.n_nz = nz;
    out.lines = (int64_t)nd;
}

// lines of [base, base + size) -> docs (offsets start at 0); the last line may lack its newline
int parse_buffer(const char *base, size_t size, trlda_docs *docs)
{
    const char *end = base + size;
    // a lone carriage return is a line end to Python's universal newlines, not to this parser
    bool lone_cr = false;
    for (const char *c = base; (c = static_cast<const char *>(std::memchr(c, '\r', (size_t)(end - c)))); ++c)
        if (c + 1 >= end || c[1] != '\n') {
            lone_cr = true;
            break;
        }
    if (lone_cr)
        return fail(TRLDA_ERR_ARG, "carriage returns without line feeds: not parsed here");
    // pieces of about equal size, cut after a newline
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    int T = (int)std::max<size_t>(1, std::min<size_t>({(size_t)hw, (size_t)64, size / ((size_t)1 << 20)}));
    if (const char *env = std::getenv("TRLDA_PARSE_THREADS"))
        T = std::max(1, std::min(std::atoi(env), 256));
    std::vector<const char *> cut((size_t)T + 1, end);
    cut[0] = base;
    for (int t = 1; t < T; ++t) {
        const char *guess = base + size / (size_t)T * (size_t)t;
        if (guess < cut[(size_t)t - 1])
            guess = cut[(size_t)t - 1];
        const char *nl = static_cast<const char *>(std::memchr(guess, '\n', (size_t)(end - guess)));
        cut[(size_t)t] = nl ? nl + 1 : end;
    }
    std::vector<TextPiece> pieces((size_t)T);
    host_pool().run(T, [&](int t) { parse_piece(cut[(size_t)t], cut[(size_t)t + 1], pieces[(size_t)t]); });
    int64_t line0 = 0;
    for (int t = 0; t < T; ++t) {
        if (pieces[(size_t)t].bad_line >= 0) {
            const int64_t line = line0 + pieces[(size_t)t].bad_line + 1;
            return fail(TRLDA_ERR_VALUE, "line " + std::to_string(line) +
                                             ": expected tokens of the form <int>:<int>");
        }
        line0 += pieces[(size_t)t].lines;
    }
    size_t ndocs = 0, nnz = 0;
    for (auto &pc : pieces) {
        ndocs += pc.n_docs;
        nnz += pc.n_nz;
    }
    docs->offsets.resize(ndocs + 1);
    docs->ids.resize(nnz);
    docs->cnts.resize(nnz);
    std::vector<size_t> doc0((size_t)T), nz0((size_t)T);
    {
        size_t d = 0, z = 0;
        for (int t = 0; t < T; ++t) {
            doc0[(size_t)t] = d;
            nz0[(size_t)t] = z;
            d += pieces[(size_t)t].n_docs;
            z += pieces[(size_t)t].n_nz;
        }
    }
    host_pool().run(T, [&](int t) {
        const TextPiece &pc = pieces[(size_t)t];
        int64_t z = (int64_t)nz0[(size_t)t];
        for (size_t i = 0; i < pc.n_docs; ++i) {
            docs->offsets[doc0[(size_t)t] + i] = z;
            z += pc.lens[i];
        }
        if (pc.n_nz) {
            std::memcpy(docs->ids.data() + nz0[(size_t)t], pc.ids.get(), pc.n_nz * 4);
            std::memcpy(docs->cnts.data() + nz0[(size_t)t], pc.cnts.get(), pc.n_nz * 4);
        }
    });
    docs->offsets[ndocs] = (int64_t)nnz;
    return TRLDA_OK;
}

}  // namespace

extern "C" {

int trlda_docs_from_text(const char *path, trlda_docs **out)
{
    if (!path || !out)
        return fail(TRLDA_ERR_ARG, "path / out is NULL");
    *out = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0)
        return fail(TRLDA_ERR_ARG, std::string("cannot open ") + path);
    struct stat st;
    if (fstat(fd, &st) != 0) {
        close(fd);
        return fail(TRLDA_ERR_ARG, std::string("cannot stat ") + path);
    }
    const size_t size = (size_t)st.st_size;
    std::unique_ptr<trlda_docs> docs(new trlda_docs());
    docs->offsets.push_back(0);
    if (size == 0) {
        close(fd);
        *out = docs.release();
        return TRLDA_OK;
    }
    void *map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED)
        return fail(TRLDA_ERR_ARG, std::string("cannot map ") + path);
    (void)madvise(map, size, MADV_SEQUENTIAL);
    const int rc = parse_buffer(static_cast<const char *>(map), size, docs.get());
    munmap(map, size);
    if (rc)
        return rc;
    *out = docs.release();
    return TRLDA_OK;
}

int trlda_docs_from_buffer(const char *text, size_t bytes, trlda_docs **out)
{
    if (!out || (!text && bytes))
        return fail(TRLDA_ERR_ARG, "text / out is NULL");
    *out = nullptr;
    std::unique_ptr<trlda_docs> docs(new trlda_docs());
    docs->offsets.push_back(0);
    if (bytes) {
        const int rc = parse_buffer(text, bytes, docs.get());
        if (rc)
            return rc;
    }
    *out = docs.release();
    return TRLDA_OK;
}

int64_t trlda_docs_num_docs(const trlda_docs *d) { return d ? (int64_t)d->offsets.size() - 1 : 0; }
int64_t trlda_docs_nnz(const trlda_docs *d) { return d ? (int64_t)d->ids.size() : 0; }
const int64_t *trlda_docs_offsets(const trlda_docs *d) { return d ? d->offsets.data() : nullptr; }
const int32_t *trlda_docs_ids(const trlda_docs *d) { return d ? d->ids.data() : nullptr; }
const int32_t *trlda_docs_cnts(const trlda_docs *d) { return d ? d->cnts.data() : nullptr; }

int trlda_docs_destroy(trlda_docs *d)
{
    delete d;
    return TRLDA_OK;
}

} // extern "C"
