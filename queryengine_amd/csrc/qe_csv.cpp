// qe_csv.cpp -- CSV text -> columns at the boundary: the on-disk step in front of the hot path.
//
// Follows the conversion rules of the reference's CSV scan leaves, data/CsvTable.kt:12-29 +
// operator/CsvSourceOperator.kt:52-76 (commons-csv 1.8, CSVFormat.DEFAULT.withFirstRecordAsHeader().withDelimiter(',')
// .withIgnoreEmptyLines(true)) and data/UnivocityCsvTable.kt:52-66 (univocity-parsers 2.8.4).  Both parsing libraries are
// un-vendored Maven dependencies that are absent from the image; what is restated here is their documented RFC 4180
// dialect as the reference configures it:
//   * records end with \n, \r\n or \r; completely empty lines are skipped; the first record is the header;
//   * fields are separated by ',', a field may be enclosed in '"' (then it may hold ',', line breaks and "" = one quote);
//   * projected fields are located by header name (first occurrence);
//   * a field missing at the end of a short record, or an empty string, is NULL (CsvSourceOperator.kt:59-62,71-73);
//   * STRING as is; BOOLEAN = String.toBoolean() = case-insensitive "true" (:65); DOUBLE = String.toDouble() =
//     java.lang.Double.parseDouble (:66): characters <= U+0020 trimmed, optional sign, decimal or hexadecimal floating
//     literal, "NaN", "Infinity", optional d/D/f/F suffix -- anything else is the reference's NumberFormatException
//     (here QE_ERR_INVALID_ARG with the offending text).
// Instead of boxing one row at a time it fills one contiguous array per column (+ validity bitmap, + a dictionary in order
// of first appearance for STRING): exactly the qe_col_desc layout that qe_batch_create pins to HBM.  Host-side I/O, once per
// table, never inside a step.
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "qe_internal.h"

struct qe_csv_table {
    int64_t nrows = 0;
    struct Col {
        int type = 0;
        std::vector<double> f64;
        std::vector<int32_t> codes;
        std::vector<uint64_t> bits;       // BOOLEAN values
        std::vector<uint64_t> validity;
        bool any_null = false;
        qe_dict dict;                     // STRING
    };
    std::vector<Col> cols;
};

namespace qe {
namespace {

inline void set_bit(std::vector<uint64_t> &v, int64_t i) {
    if ((size_t)(i >> 6) >= v.size()) v.resize((size_t)(i >> 6) + 1, 0);
    v[(size_t)(i >> 6)] |= 1ull << (i & 63);
}

// java.lang.Double.parseDouble; returns false for what Java rejects
bool java_parse_double(const char *p, size_t n, double &out) {
    while (n > 0 && (unsigned char)p[0] <= 0x20) { p++; n--; }
    while (n > 0 && (unsigned char)p[n - 1] <= 0x20) n--;
    if (n == 0) return false;
    std::string s(p, n);
    size_t i = 0;
    bool neg = false;
    if (s[i] == '+' || s[i] == '-') { neg = s[i] == '-'; i++; }
    const std::string body = s.substr(i);
    if (body == "NaN") { out = std::nan(""); return true; }
    if (body == "Infinity") { out = neg ? -INFINITY : INFINITY; return true; }
    // validate the literal's shape (strtod is more liberal than Java: "inf", "nan(..)", "0x10" without exponent ...)
    size_t j = 0;
    const size_t m = body.size();
    auto digits = [&](bool hex) {
        size_t k = 0;
        while (j < m && (hex ? isxdigit((unsigned char)body[j]) : isdigit((unsigned char)body[j]))) { j++; k++; }
        return k;
    };
    bool hex = m > 2 && body[0] == '0' && (body[1] == 'x' || body[1] == 'X');
    if (hex) {
        j = 2;
        size_t a = digits(true), b = 0;
        if (j < m && body[j] == '.') { j++; b = digits(true); }
        if (a + b == 0) return false;
        if (j >= m || (body[j] != 'p' && body[j] != 'P')) return false;   // the binary exponent is mandatory
        j++;
        if (j < m && (body[j] == '+' || body[j] == '-')) j++;
        if (digits(false) == 0) return false;
    } else {
        size_t a = digits(false), b = 0;
        if (j < m && body[j] == '.') { j++; b = digits(false); }
        if (a + b == 0) return false;
        if (j < m && (body[j] == 'e' || body[j] == 'E')) {
            j++;
            if (j < m && (body[j] == '+' || body[j] == '-')) j++;
            if (digits(false) == 0) return false;
        }
    }
    size_t end = j;
    if (j < m && (body[j] == 'd' || body[j] == 'D' || body[j] == 'f' || body[j] == 'F')) j++;
    if (j != m) return false;
    const std::string lit = (neg ? "-" : "") + body.substr(0, end);
    char *ep = nullptr;
    errno = 0;
    out = std::strtod(lit.c_str(), &ep);   // correctly rounded (glibc), like Java; over/underflow give inf / 0 as Java does
    return ep && *ep == '\0';
}

bool kotlin_to_boolean(const char *p, size_t n) {
    return n == 4 && (p[0] | 0x20) == 't' && (p[1] | 0x20) == 'r' && (p[2] | 0x20) == 'u' && (p[3] | 0x20) == 'e';
}

// One record of the RFC 4180 dialect above.  `pos` advances past the record's line end.  Returns false at end of input
// (no record).  `empty_line` is set for a completely empty line.
struct Field { size_t begin, end; bool quoted; };
bool next_record(const char *d, size_t n, size_t &pos, std::vector<Field> &fields, std::string &unq, std::vector<std::pair<size_t, size_t>> &unq_span,
                 bool &empty_line) {
    fields.clear();
    unq.clear();
    unq_span.clear();
    empty_line = false;
    if (pos >= n) return false;
    if (d[pos] == '\n' || d[pos] == '\r') {   // an empty line
        if (d[pos] == '\r' && pos + 1 < n && d[pos + 1] == '\n') pos++;
        pos++;
        empty_line = true;
        return true;
    }
    for (;;) {
        Field f{pos, pos, false};
        if (pos < n && d[pos] == '"') {   // enclosed field: copy with "" -> "
            f.quoted = true;
            pos++;
            const size_t ub = unq.size();
            for (;;) {
                if (pos >= n) fail(QE_ERR_INVALID_ARG, "CSV: end of input inside a quoted field");
                if (d[pos] == '"') {
                    if (pos + 1 < n && d[pos + 1] == '"') { unq.push_back('"'); pos += 2; continue; }
                    pos++;
                    break;
                }
                unq.push_back(d[pos++]);
            }
            unq_span.emplace_back(ub, unq.size());
            f.begin = unq_span.size() - 1;   // index into unq_span
            if (pos < n && d[pos] != ',' && d[pos] != '\n' && d[pos] != '\r')
                fail(QE_ERR_INVALID_ARG, "CSV: invalid character after a closing quote");
        } else {
            while (pos < n && d[pos] != ',' && d[pos] != '\n' && d[pos] != '\r') pos++;
            f.end = pos;
        }
        fields.push_back(f);
        if (pos >= n) return true;
        if (d[pos] == ',') { pos++; if (pos >= n) { fields.push_back(Field{pos, pos, false}); return true; } continue; }
        if (d[pos] == '\r' && pos + 1 < n && d[pos + 1] == '\n') pos++;
        pos++;
        return true;
    }
}

}  // namespace
}  // namespace qe

using namespace qe;

template <typename F>
static int32_t guarded_csv(qe_ctx *ctx, F &&f) {
    try {
        f();
        return QE_OK;
    } catch (const Error &e) {
        if (ctx) ctx->last_error = e.msg;
        return e.code;
    } catch (const std::bad_alloc &) {
        if (ctx) ctx->last_error = "host out of memory";
        return QE_ERR_OOM;
    } catch (const std::exception &e) {
        if (ctx) ctx->last_error = e.what();
        return QE_ERR_INTERNAL;
    }
}

extern "C" {

int32_t qe_csv_parse(qe_ctx *ctx, const char *data, size_t nbytes, int32_t nfields, const char *const *names, const int32_t *types,
                     qe_csv_table **out) {
    if (!ctx || !out || (!data && nbytes) || nfields < 0 || (nfields > 0 && (!names || !types))) return QE_ERR_INVALID_ARG;
    *out = nullptr;
    return guarded_csv(ctx, [&] {
        std::unique_ptr<qe_csv_table> t(new qe_csv_table());
        t->cols.resize((size_t)nfields);
        for (int32_t k = 0; k < nfields; k++) {
            if (!names[k]) fail(QE_ERR_INVALID_ARG, "null field name");
            if (types[k] != QE_STRING && types[k] != QE_DOUBLE && types[k] != QE_BOOLEAN)
                fail(QE_ERR_INVALID_ARG, "CSV sources carry the reference's three types only (STRING, DOUBLE, BOOLEAN)");
            t->cols[(size_t)k].type = types[k];
            if (types[k] == QE_STRING) t->cols[(size_t)k].dict.d = std::make_shared<DictData>();
        }
        // (a UTF-8 byte order mark stays part of the first header name, as with the reference's FileReader(file, UTF_8))
        std::vector<Field> rec;
        std::string unq;
        std::vector<std::pair<size_t, size_t>> span;
        size_t pos = 0;
        bool empty = false;
        auto text = [&](const Field &f, const char *&p, size_t &n) {
            if (f.quoted) { p = unq.data() + span[f.begin].first; n = span[f.begin].second - span[f.begin].first; }
            else { p = data + f.begin; n = f.end - f.begin; }
        };
        // header: the first non-empty record (withFirstRecordAsHeader + withIgnoreEmptyLines)
        std::vector<int> idx((size_t)nfields, -1);
        bool have_header = false;
        while (next_record(data, nbytes, pos, rec, unq, span, empty)) {
            if (empty) continue;
            have_header = true;
            for (int32_t k = 0; k < nfields; k++) {
                for (size_t i = 0; i < rec.size(); i++) {
                    const char *p; size_t n;
                    text(rec[i], p, n);
                    // a duplicated header name resolves to its LAST occurrence: commons-csv 1.8 (CSVFormat.DEFAULT allows
                    // duplicates) fills its header map with put(), so a later column replaces an earlier one of the same name
                    if (n == std::strlen(names[k]) && std::memcmp(p, names[k], n) == 0) idx[(size_t)k] = (int)i;
                }
            }
            break;
        }
        for (int32_t k = 0; k < nfields; k++)
            if (idx[(size_t)k] < 0)   // CsvSourceOperator.kt:27-28
                fail(QE_ERR_INVALID_ARG, std::string("projected field ") + names[k] + " not found in csv headers");
        (void)have_header;
        int64_t row = 0;
        while (next_record(data, nbytes, pos, rec, unq, span, empty)) {
            if (empty) continue;
            for (int32_t k = 0; k < nfields; k++) {
                qe_csv_table::Col &c = t->cols[(size_t)k];
                const char *p = nullptr;
                size_t n = 0;
                if ((size_t)idx[(size_t)k] < rec.size()) text(rec[(size_t)idx[(size_t)k]], p, n);
                const bool is_null = n == 0;   // missing trailing field or isNullOrEmpty (:59-62)
                if (c.type == QE_DOUBLE) {
                    double v = 0.0;
                    if (!is_null && !java_parse_double(p, n, v))
                        fail(QE_ERR_INVALID_ARG, "NumberFormatException: For input string: \"" + std::string(p, n) + "\" (row " +
                                                     std::to_string(row + 1) + ", field " + names[k] + ")");
                    c.f64.push_back(v);
                } else if (c.type == QE_BOOLEAN) {
                    if ((size_t)(row >> 6) >= c.bits.size()) c.bits.resize((size_t)(row >> 6) + 1, 0);
                    if (!is_null && kotlin_to_boolean(p, n)) set_bit(c.bits, row);
                } else {
                    int32_t code = 0;
                    if (!is_null) {
                        std::string s(p, n);
                        auto it = c.dict.d->index.find(s);
                        if (it == c.dict.d->index.end()) {
                            code = (int32_t)c.dict.d->entries.size();
                            c.dict.d->entries.push_back(s);
                            c.dict.d->index.emplace(std::move(s), code);
                        } else {
                            code = it->second;
                        }
                    }
                    c.codes.push_back(code);
                }
                if ((size_t)(row >> 6) >= c.validity.size()) c.validity.resize((size_t)(row >> 6) + 1, 0);
                if (is_null) c.any_null = true;
                else set_bit(c.validity, row);
            }
            row++;
        }
        t->nrows = row;
        for (auto &c : t->cols) {
            const size_t words = (size_t)((row + 63) / 64);
            c.validity.resize(std::max<size_t>(words, 1), 0);
            if (c.type == QE_BOOLEAN) c.bits.resize(std::max<size_t>(words, 1), 0);
            if (c.type == QE_DOUBLE && c.f64.empty()) c.f64.push_back(0.0);
            if (c.type == QE_STRING && c.codes.empty()) c.codes.push_back(0);
        }
        *out = t.release();
    });
}

int32_t qe_csv_parse_file(qe_ctx *ctx, const char *path, int32_t nfields, const char *const *names, const int32_t *types, qe_csv_table **out) {
    if (!ctx || !path || !out) return QE_ERR_INVALID_ARG;
    *out = nullptr;
    std::vector<char> buf;
    int32_t st = guarded_csv(ctx, [&] {
        FILE *f = std::fopen(path, "rb");
        if (!f) fail(QE_ERR_INVALID_ARG, std::string("cannot open ") + path + ": " + std::strerror(errno));
        struct Closer { FILE *f; ~Closer() { std::fclose(f); } } closer{f};
        char chunk[1 << 16];
        size_t n;
        while ((n = std::fread(chunk, 1, sizeof chunk, f)) > 0) buf.insert(buf.end(), chunk, chunk + n);
        if (std::ferror(f)) fail(QE_ERR_INVALID_ARG, std::string("read error on ") + path);
    });
    if (st != QE_OK) return st;
    return qe_csv_parse(ctx, buf.data(), buf.size(), nfields, names, types, out);
}

int64_t qe_csv_nrows(const qe_csv_table *t) { return t ? t->nrows : -1; }
int32_t qe_csv_ncols(const qe_csv_table *t) { return t ? (int32_t)t->cols.size() : -1; }

// host-side column in the qe_col_desc layout (pointers owned by the table; validity == NULL when the column has no NULL)
int32_t qe_csv_column(const qe_csv_table *t, int32_t col, qe_col_desc *out) {
    if (!t || !out || col < 0 || col >= (int32_t)t->cols.size()) return QE_ERR_INVALID_ARG;
    const qe_csv_table::Col &c = t->cols[(size_t)col];
    out->type = c.type;
    out->reserved = 0;
    out->data = c.type == QE_DOUBLE ? (const void *)c.f64.data() : c.type == QE_BOOLEAN ? (const void *)c.bits.data() : (const void *)c.codes.data();
    out->validity = c.any_null ? c.validity.data() : nullptr;
    out->dict = c.type == QE_STRING ? &c.dict : nullptr;
    return QE_OK;
}

// pin the parsed columns to HBM: qe_batch_create on the table's own buffers
int32_t qe_csv_pin(qe_ctx *ctx, const qe_csv_table *t, qe_batch **out) {
    if (!ctx || !t || !out) return QE_ERR_INVALID_ARG;
    std::vector<qe_col_desc> descs(t->cols.size());
    for (size_t j = 0; j < t->cols.size(); j++) (void)qe_csv_column(t, (int32_t)j, &descs[j]);
    return qe_batch_create(ctx, t->nrows, (int32_t)descs.size(), descs.data(), out);
}

void qe_csv_free(qe_ctx *, qe_csv_table *t) { delete t; }

}  // extern "C"
