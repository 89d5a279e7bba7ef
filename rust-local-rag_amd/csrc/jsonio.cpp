// jsonio.cpp -- the corpus file `chunks_{model}.json` at scale: a streaming reader that
// pulls the embedding arrays out of the reference's PersistedState document (src/rag_engine.rs:1478-1499, :1525-1535)
// straight into a dense f32 matrix, and the matching number formatter for the writer.  Host code only.
//
// The reference parses the whole pretty-printed file with serde_json (:1555-1557); at 10^5..10^6 chunks the
// `Vec<f32>` literals are > 99 % of the bytes.  Here the file is memory-mapped and read in two passes:
//   1. structure: a small JSON tokenizer walks the document once, tracking where it is (top level -> "chunks" ->
//      chunk object -> "embedding"); an embedding array holds no strings, so its end is one memchr for ']' -- the
//      pass only records the byte range of every array (GB/s) and copies everything else verbatim into a "metadata
//      document" in which each embedding array is replaced by [], small enough for any JSON library;
//   2. numbers: the recorded ranges are parsed by all host cores at once (std::from_chars<double>, correctly
//      rounded, then narrowed to binary32 -- serde_json's f32 path (visit_f64 + `as f32`) and the Python loader's
//      float64 -> float32 do exactly that, so the rows are bit-identical to theirs), each range into its own row of
//      the matrix: about 25 ns per number per core.
// Row r = the r-th chunk of the file.
#include "../../include/rlr_engine.h"

#include <algorithm>
#include <atomic>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace rlr {
int32_t set_error(int32_t code, const char *fmt, ...); // index.hip: the calling thread's rlr_last_error() text
}

namespace {

struct Cursor {
    const char *p, *end, *begin;
    std::string err;
    bool fail(const char *what)
    {
        if (err.empty()) {
            char b[160];
            snprintf(b, sizeof b, "%s at byte %zu", what, static_cast<size_t>(p - begin));
            err = b;
        }
        return false;
    }
    void ws()
    {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r'))
            ++p;
    }
};

// p at the opening quote; leaves p behind the closing quote; [s, e) = raw bytes between the quotes
bool skip_string(Cursor &c, const char **s, const char **e)
{
    if (c.p >= c.end || *c.p != '"')
        return c.fail("expected a string");
    ++c.p;
    *s = c.p;
    while (c.p < c.end && *c.p != '"') {
        if (*c.p == '\\') {
            ++c.p;
            if (c.p >= c.end)
                break;
        }
        ++c.p;
    }
    if (c.p >= c.end)
        return c.fail("unterminated string");
    *e = c.p;
    ++c.p;
    return true;
}

bool skip_value(Cursor &c);

bool skip_container(Cursor &c, char open, char close)
{
    int depth = 0;
    while (c.p < c.end) {
        const char ch = *c.p;
        if (ch == '"') {
            const char *s, *e;
            if (!skip_string(c, &s, &e))
                return false;
            continue;
        }
        if (ch == '{' || ch == '[')
            ++depth;
        else if (ch == '}' || ch == ']') {
            --depth;
            if (depth == 0) {
                if (ch != close)
                    return c.fail("mismatched bracket");
                ++c.p;
                return true;
            }
        }
        ++c.p;
    }
    (void)open;
    return c.fail("unterminated container");
}

bool skip_value(Cursor &c)
{
    c.ws();
    if (c.p >= c.end)
        return c.fail("unexpected end of input");
    const char ch = *c.p;
    if (ch == '"') {
        const char *s, *e;
        return skip_string(c, &s, &e);
    }
    if (ch == '{')
        return skip_container(c, '{', '}');
    if (ch == '[')
        return skip_container(c, '[', ']');
    const char *s = c.p;
    while (c.p < c.end && *c.p != ',' && *c.p != '}' && *c.p != ']' && *c.p != ' ' && *c.p != '\n' && *c.p != '\t' &&
           *c.p != '\r')
        ++c.p;
    if (c.p == s)
        return c.fail("expected a value");
    return true;
}

bool key_is(const char *s, const char *e, const char *lit)
{
    const size_t n = strlen(lit);
    return static_cast<size_t>(e - s) == n && memcmp(s, lit, n) == 0;
}

// [p, end) = the inside of an embedding array (between '[' and ']'): numbers (or null = a non-finite value serde_json
// wrote) into row[0..dim), extra components dropped, missing ones left 0 (dot_product's zip, rag_engine.rs:1778).
// Returns nullptr, or the position of the first byte that is not part of a number list.
const char *parse_embedding(const char *p, const char *end, float *row, uint32_t dim)
{
    uint32_t i = 0;
    for (;;) {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r' || *p == ','))
            ++p;
        if (p >= end)
            return nullptr;
        double v;
        if (end - p >= 4 && memcmp(p, "null", 4) == 0) {
            v = std::nan("");
            p += 4;
        } else {
            const auto r = std::from_chars(p, end, v);
            if (r.ec == std::errc::result_out_of_range) {
                // from_chars leaves v unmodified on overflow / underflow: serde_json's f64 gives +-inf or 0 there
                const bool neg = *p == '-';
                v = 0.0;
                for (const char *t = p; t < r.ptr; ++t)
                    if (*t == 'e' || *t == 'E') {
                        v = (t + 1 < r.ptr && t[1] == '-') ? 0.0 : HUGE_VAL;
                        break;
                    }
                if (neg)
                    v = -v;
            } else if (r.ec != std::errc()) {
                return p;
            }
            p = r.ptr;
        }
        if (i < dim)
            row[i] = static_cast<float>(v);
        ++i;
    }
}

struct Range {
    const char *b, *e; // inside of the array
    uint64_t row;
};

struct Corpus {
    std::vector<Range> ranges; // one per embedding array found (a chunk without one keeps its zero row)
    uint64_t n = 0;
    std::string meta;
};

bool parse_document(Cursor &c, Corpus &out)
{
    const char *last = c.begin; // everything in [last, p) still has to be copied to the metadata document
    c.ws();
    if (c.p >= c.end || *c.p != '{')
        return c.fail("the document is not a JSON object");
    ++c.p;
    for (;;) { // top-level members
        c.ws();
        if (c.p < c.end && *c.p == '}') {
            ++c.p;
            break;
        }
        if (c.p < c.end && *c.p == ',') {
            ++c.p;
            continue;
        }
        const char *ks, *ke;
        if (!skip_string(c, &ks, &ke))
            return false;
        c.ws();
        if (c.p >= c.end || *c.p != ':')
            return c.fail("expected ':'");
        ++c.p;
        c.ws();
        if (!key_is(ks, ke, "chunks") || c.p >= c.end || *c.p != '{') {
            if (!skip_value(c))
                return false;
            continue;
        }
        ++c.p; // '{' of the chunk map
        for (;;) {
            c.ws();
            if (c.p < c.end && *c.p == '}') {
                ++c.p;
                break;
            }
            if (c.p < c.end && *c.p == ',') {
                ++c.p;
                continue;
            }
            if (!skip_string(c, &ks, &ke)) // chunk id
                return false;
            c.ws();
            if (c.p >= c.end || *c.p != ':')
                return c.fail("expected ':'");
            ++c.p;
            c.ws();
            // one row per chunk, whatever the chunk holds
            const uint64_t row = out.n++;
            size_t first_range = out.ranges.size();
            if (c.p >= c.end || *c.p != '{') {
                if (!skip_value(c))
                    return false;
                continue;
            }
            ++c.p;
            for (;;) { // chunk members
                c.ws();
                if (c.p < c.end && *c.p == '}') {
                    ++c.p;
                    break;
                }
                if (c.p < c.end && *c.p == ',') {
                    ++c.p;
                    continue;
                }
                if (!skip_string(c, &ks, &ke))
                    return false;
                c.ws();
                if (c.p >= c.end || *c.p != ':')
                    return c.fail("expected ':'");
                ++c.p;
                c.ws();
                if (key_is(ks, ke, "embedding") && c.p < c.end && *c.p == '[') {
                    const char *a0 = c.p;
                    // numbers, commas, white space and null only: the array ends at the next ']'
                    const char *close = static_cast<const char *>(memchr(c.p + 1, ']', static_cast<size_t>(c.end - c.p - 1)));
                    if (!close)
                        return c.fail("unterminated embedding array");
                    out.ranges.resize(first_range); // a repeated key: the last one wins, as in a map
                    out.ranges.push_back({c.p + 1, close, row});
                    c.p = close + 1;
                    out.meta.append(last, a0);
                    out.meta.append("[]");
                    last = c.p;
                } else if (!skip_value(c)) {
                    return false;
                }
            }
        }
    }
    out.meta.append(last, c.end);
    return true;
}

// pass 2: every recorded array into its row, all host cores (bounded), work handed out in blocks of 64 arrays
const char *parse_ranges(const std::vector<Range> &ranges, float *rows, uint32_t dim)
{
    const size_t n = ranges.size();
    unsigned n_thr = std::thread::hardware_concurrency();
    n_thr = std::max(1u, std::min(n_thr ? n_thr : 1u, 32u));
    if (n < 256)
        n_thr = 1;
    std::atomic<size_t> next{0};
    std::atomic<const char *> bad{nullptr};
    auto work = [&] {
        for (;;) {
            const size_t i0 = next.fetch_add(64);
            if (i0 >= n || bad.load(std::memory_order_relaxed))
                return;
            for (size_t i = i0; i < std::min(n, i0 + 64); ++i) {
                const char *err = parse_embedding(ranges[i].b, ranges[i].e, rows + ranges[i].row * dim, dim);
                if (err) {
                    bad.store(err);
                    return;
                }
            }
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < n_thr; ++t)
        th.emplace_back(work);
    work();
    for (auto &t : th)
        t.join();
    return bad.load();
}

struct Mapped {
    void *map = MAP_FAILED;
    size_t size = 0;
    ~Mapped()
    {
        if (map != MAP_FAILED)
            munmap(map, size);
    }
};

// rows_out: malloc'ed n x dim floats (zero rows for chunks without an embedding)
int32_t load_corpus(const char *path, uint32_t dim, Corpus &out, float **rows_out)
{
    *rows_out = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0)
        return rlr::set_error(RLR_E_INVALID, "cannot open %s", path);
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size == 0) {
        close(fd);
        return rlr::set_error(RLR_E_INVALID, "%s is empty or unreadable", path);
    }
    Mapped m;
    m.size = static_cast<size_t>(st.st_size);
    m.map = mmap(nullptr, m.size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m.map == MAP_FAILED)
        return rlr::set_error(RLR_E_OOM, "cannot map %s", path);
    (void)madvise(m.map, m.size, MADV_WILLNEED);
    Cursor c;
    c.begin = c.p = static_cast<const char *>(m.map);
    c.end = c.p + m.size;
    if (!parse_document(c, out))
        return rlr::set_error(RLR_E_INVALID, "%s: %s", path, c.err.c_str());
    float *rows = static_cast<float *>(std::calloc(std::max<size_t>(out.n * static_cast<size_t>(dim), 1), sizeof(float)));
    if (!rows)
        return rlr::set_error(RLR_E_OOM, "host allocation of %llu x %u floats failed", static_cast<unsigned long long>(out.n), dim);
    if (const char *err = parse_ranges(out.ranges, rows, dim)) {
        std::free(rows);
        return rlr::set_error(RLR_E_INVALID, "%s: bad number in an embedding array at byte %zu", path,
                              static_cast<size_t>(err - c.begin));
    }
    *rows_out = rows;
    return RLR_OK;
}

} // namespace

extern "C" {

int32_t rlr_json_load_corpus(const char *path, uint32_t dim, rlr_json_corpus *out)
{
    if (!path || !out || dim == 0)
        return RLR_E_INVALID;
    std::memset(out, 0, sizeof(*out));
    Corpus c;
    float *rows = nullptr;
    const int32_t st = load_corpus(path, dim, c, &rows);
    if (st != RLR_OK)
        return st;
    out->n_rows = c.n;
    out->dim = dim;
    out->meta_len = c.meta.size();
    out->rows = rows;
    out->meta_json = static_cast<char *>(std::malloc(c.meta.size() + 1));
    if (!out->meta_json) {
        rlr_json_free_corpus(out);
        return rlr::set_error(RLR_E_OOM, "host allocation failed");
    }
    std::memcpy(out->meta_json, c.meta.data(), c.meta.size());
    out->meta_json[c.meta.size()] = '\0';
    return RLR_OK;
}

void rlr_json_free_corpus(rlr_json_corpus *c)
{
    if (!c)
        return;
    std::free(c->rows);
    std::free(c->meta_json);
    std::memset(c, 0, sizeof(*c));
}

int32_t rlr_index_load_json(rlr_index *idx, const char *path, int32_t normalize_on_device, rlr_json_corpus *meta_out)
{
    if (!idx || !path)
        return RLR_E_INVALID;
    uint32_t dim = 0;
    int32_t st = rlr_index_info(idx, nullptr, &dim, nullptr, nullptr);
    if (st != RLR_OK)
        return st;
    rlr_json_corpus c;
    st = rlr_json_load_corpus(path, dim, &c);
    if (st != RLR_OK)
        return st;
    st = rlr_index_upload(idx, c.rows, c.n_rows, normalize_on_device);
    if (st != RLR_OK || !meta_out) {
        rlr_json_free_corpus(&c);
        return st;
    }
    std::free(c.rows); // the rows live in HBM now; the caller keeps the metadata document
    c.rows = nullptr;
    *meta_out = c;
    return RLR_OK;
}

// "[\n<indent + 2 spaces>v0,\n ... \n<indent>]" with every value the shortest decimal that reads back as the same
// binary32 (what serde_json / Ryu print), non-finite values as null (serde_json does the same).  Returns the bytes
// the text needs; it was written (without a terminating NUL) only if that is <= cap.
uint64_t rlr_json_format_embedding(const float *v, uint32_t dim, uint32_t indent, char *out, uint64_t cap)
{
    std::string s;
    s.reserve(static_cast<size_t>(dim) * (indent + 16) + 8);
    if (dim == 0) {
        s = "[]";
    } else {
        s = "[\n";
        char buf[64];
        for (uint32_t i = 0; i < dim; ++i) {
            s.append(indent + 2, ' ');
            if (!std::isfinite(v[i])) {
                s.append("null");
            } else {
                const auto r = std::to_chars(buf, buf + sizeof buf, v[i]);
                bool plain = true;
                for (const char *t = buf; t < r.ptr; ++t)
                    if (*t == '.' || *t == 'e' || *t == 'E')
                        plain = false;
                s.append(buf, r.ptr);
                if (plain)
                    s.append(".0");
            }
            s.append(i + 1 < dim ? ",\n" : "\n");
        }
        s.append(indent, ' ');
        s.push_back(']');
    }
    if (out && s.size() <= cap)
        std::memcpy(out, s.data(), s.size());
    return s.size();
}

} // extern "C"
