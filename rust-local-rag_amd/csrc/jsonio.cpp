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
// Row r = the r-th DISTINCT chunk id of the file's (last) "chunks" map, in order of first appearance.
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
#include <unordered_map>
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

// The JSON number grammar, -?(0|[1-9][0-9]*)(\.[0-9]+)?([eE][+-]?[0-9]+)?, as serde_json and Python's json module
// enforce it (std::from_chars alone also takes "inf", "nan", "01", "1.", ".5").  Returns the end of the literal or
// nullptr; *mag10 = decimal order of magnitude of the value (position of its first significant digit plus the exponent,
// saturated) -- what decides between +-inf and +-0 when the literal is outside binary64's range.
const char *scan_json_number(const char *p, const char *end, long *mag10)
{
    const char *q = p;
    if (q < end && *q == '-')
        ++q;
    if (q >= end || *q < '0' || *q > '9')
        return nullptr;
    long first_sig = 0; // 10^first_sig is the weight of the first non-zero digit
    bool seen_sig = false;
    const char *int_begin = q;
    if (*q == '0') {
        ++q;
    } else {
        while (q < end && *q >= '0' && *q <= '9')
            ++q;
        seen_sig = true;
        first_sig = static_cast<long>(q - int_begin) - 1;
    }
    if (q < end && *q == '.') {
        ++q;
        const char *frac = q;
        while (q < end && *q >= '0' && *q <= '9') {
            if (!seen_sig && *q != '0') {
                seen_sig = true;
                first_sig = -static_cast<long>(q - frac) - 1;
            }
            ++q;
        }
        if (q == frac)
            return nullptr;
    }
    long ex = 0;
    if (q < end && (*q == 'e' || *q == 'E')) {
        ++q;
        bool neg = false;
        if (q < end && (*q == '+' || *q == '-'))
            neg = *q++ == '-';
        const char *d = q;
        while (q < end && *q >= '0' && *q <= '9') {
            if (ex < 100000000)
                ex = ex * 10 + (*q - '0');
            ++q;
        }
        if (q == d)
            return nullptr;
        if (neg)
            ex = -ex;
    }
    *mag10 = seen_sig ? first_sig + ex : 0;
    return q;
}

// [p, end) = the inside of an embedding array (between '[' and ']'): `value (, value)*` or nothing, a value being a JSON
// number or null (= a non-finite value serde_json wrote), into row[0..dim); extra components dropped, missing ones left 0
// (dot_product's zip, rag_engine.rs:1778).  Returns nullptr, or the position of the first byte that breaks that grammar
// (a nested array, a string, "inf", a doubled or trailing comma, ...: serde_json refuses such a file, :1555-1557).
const char *parse_embedding(const char *p, const char *end, float *row, uint32_t dim)
{
    auto ws = [&] {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r'))
            ++p;
    };
    uint32_t i = 0;
    ws();
    if (p >= end)
        return nullptr; // []
    for (;;) {
        double v;
        if (end - p >= 4 && memcmp(p, "null", 4) == 0) {
            v = std::nan("");
            p += 4;
        } else {
            long mag10 = 0;
            const char *lit_end = scan_json_number(p, end, &mag10);
            if (!lit_end)
                return p;
            const auto r = std::from_chars(p, lit_end, v);
            if (r.ec == std::errc::result_out_of_range) {
                // from_chars leaves v unmodified on overflow / underflow: serde_json's f64 gives +-inf or +-0 there
                v = mag10 > 0 ? HUGE_VAL : 0.0;
                if (*p == '-')
                    v = -v;
            } else if (r.ec != std::errc() || r.ptr != lit_end) {
                return p;
            }
            p = lit_end;
        }
        if (i < dim)
            row[i] = static_cast<float>(v);
        ++i;
        ws();
        if (p >= end)
            return nullptr;
        if (*p != ',')
            return p;
        ++p;
        ws();
        if (p >= end)
            return p - 1; // trailing comma
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

constexpr uint64_t kDeadRange = ~0ull; // a range whose chunk was replaced by a later occurrence of the same id

// JSON string contents [s, e) decoded (escapes, \uXXXX incl. surrogate pairs -> UTF-8): two spellings of one key must
// compare equal, as they do in serde_json's HashMap and in Python's dict.  Invalid escapes are kept as written.
std::string decode_string(const char *s, const char *e)
{
    std::string out;
    out.reserve(static_cast<size_t>(e - s));
    auto hex4 = [&](const char *q, unsigned *v) {
        if (e - q < 4)
            return false;
        unsigned x = 0;
        for (int i = 0; i < 4; ++i) {
            const char ch = q[i];
            unsigned d;
            if (ch >= '0' && ch <= '9') d = static_cast<unsigned>(ch - '0');
            else if (ch >= 'a' && ch <= 'f') d = static_cast<unsigned>(ch - 'a' + 10);
            else if (ch >= 'A' && ch <= 'F') d = static_cast<unsigned>(ch - 'A' + 10);
            else return false;
            x = x * 16 + d;
        }
        *v = x;
        return true;
    };
    auto utf8 = [&](unsigned cp) {
        if (cp < 0x80) out += static_cast<char>(cp);
        else if (cp < 0x800) {
            out += static_cast<char>(0xC0 | (cp >> 6));
            out += static_cast<char>(0x80 | (cp & 0x3F));
        } else if (cp < 0x10000) {
            out += static_cast<char>(0xE0 | (cp >> 12));
            out += static_cast<char>(0x80 | ((cp >> 6) & 0x3F));
            out += static_cast<char>(0x80 | (cp & 0x3F));
        } else {
            out += static_cast<char>(0xF0 | (cp >> 18));
            out += static_cast<char>(0x80 | ((cp >> 12) & 0x3F));
            out += static_cast<char>(0x80 | ((cp >> 6) & 0x3F));
            out += static_cast<char>(0x80 | (cp & 0x3F));
        }
    };
    for (const char *q = s; q < e; ++q) {
        if (*q != '\\' || q + 1 >= e) {
            out += *q;
            continue;
        }
        ++q;
        switch (*q) {
        case 'b': out += '\b'; break;
        case 'f': out += '\f'; break;
        case 'n': out += '\n'; break;
        case 'r': out += '\r'; break;
        case 't': out += '\t'; break;
        case 'u': {
            unsigned hi = 0, lo = 0;
            if (!hex4(q + 1, &hi)) {
                out += "\\u";
                break;
            }
            q += 4;
            if (hi >= 0xD800 && hi < 0xDC00 && e - q >= 7 && q[1] == '\\' && q[2] == 'u' && hex4(q + 3, &lo) && lo >= 0xDC00 &&
                lo < 0xE000) {
                q += 6;
                utf8(0x10000 + ((hi - 0xD800) << 10) + (lo - 0xDC00));
            } else {
                utf8(hi);
            }
            break;
        }
        default: out += *q; break; // \" \\ \/
        }
    }
    return out;
}

bool key_equals(const char *s, const char *e, const char *lit)
{
    if (memchr(s, '\\', static_cast<size_t>(e - s)) == nullptr)
        return key_is(s, e, lit);
    return decode_string(s, e) == lit;
}

bool parse_document(Cursor &c, Corpus &out)
{
    const char *last = c.begin; // everything in [last, p) still has to be copied to the metadata document
    // chunk id -> row and row -> its embedding range: the document is a MAP (serde_json HashMap / Python dict: a repeated
    // key keeps its first position and takes the last value), so a repeated id reuses its row and replaces the whole
    // chunk, and a repeated "chunks" member starts over
    std::unordered_map<std::string, uint64_t> row_of;
    std::vector<int64_t> range_of_row;
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
        const bool is_chunks = key_equals(ks, ke, "chunks");
        if (is_chunks) { // (again): the last "chunks" member is the one a map keeps
            out.n = 0;
            out.ranges.clear();
            row_of.clear();
            range_of_row.clear();
        }
        if (!is_chunks || c.p >= c.end || *c.p != '{') {
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
            // one row per chunk ID, whatever the chunk holds
            auto ins = row_of.emplace(decode_string(ks, ke), out.n);
            const uint64_t row = ins.first->second;
            if (ins.second) {
                out.n++;
                range_of_row.push_back(-1);
            } else if (range_of_row[row] >= 0) { // the earlier occurrence's embedding goes with the earlier value
                out.ranges[static_cast<size_t>(range_of_row[row])].row = kDeadRange;
                range_of_row[row] = -1;
            }
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
                const bool is_emb = key_equals(ks, ke, "embedding");
                if (is_emb && range_of_row[row] >= 0) { // a repeated key: the last one wins, as in a map
                    out.ranges[static_cast<size_t>(range_of_row[row])].row = kDeadRange;
                    range_of_row[row] = -1;
                }
                if (is_emb && c.p < c.end && *c.p == '[') {
                    const char *a0 = c.p;
                    // numbers, commas, white space and null only: the array ends at the next ']'
                    const char *close = static_cast<const char *>(memchr(c.p + 1, ']', static_cast<size_t>(c.end - c.p - 1)));
                    if (!close)
                        return c.fail("unterminated embedding array");
                    range_of_row[row] = static_cast<int64_t>(out.ranges.size());
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
    out.ranges.erase(std::remove_if(out.ranges.begin(), out.ranges.end(), [](const Range &r) { return r.row == kDeadRange; }),
                     out.ranges.end());
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
