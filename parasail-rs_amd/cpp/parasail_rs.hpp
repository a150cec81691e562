// parasail_rs.hpp -- header-only C++ mirror of the parasail-rs interface over libparasail_amd.so.
//
// The reference's host layer is Rust (/root/reference/src/{aligner,matrix,profile,alignment}),
// which cannot be compiled in this image; this header mirrors it 1:1 in C++ (same type and method
// names, argument meaning, defaults, quirks and error behaviour) on top of the C ABI in
// include/parasail_amd.h.  Rust `Result<T, Error>` becomes a thrown `parasail_rs::Error` with the
// enum variant in `.kind`; Rust panics become `parasail_rs::Panic`.
//   Rust                                   here
//   Aligner::new().local().build()         Aligner::builder().local().build()
//   aligner.align(Some(q), r)?             aligner.align(&q, r)
//   aligner.align(None, r)?                aligner.align(nullptr, r)
//   aligner.align_batch(...)               additive, no reference counterpart
#pragma once
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/parasail_amd.h"

namespace parasail_rs {

using Bytes = std::string;   // byte sequences (may not contain NUL, like CString::new)

enum class ErrorKind {           // src/error.rs + src/*/error.rs
    InteriorNulByte, NoBandwidth, NoStats, NoTable, NoStatsTable, NoRowCol, NoTrace, InvalidUTF8String,
    FailedLookup, NullMatrix, NotSquare, NotBuiltIn, InvalidIndex, FileNotFound, NullProfile, QueryIsEmpty, Batch
};
struct Error : std::runtime_error {
    ErrorKind kind;
    Error(ErrorKind k, const std::string &what) : std::runtime_error(what), kind(k) {}
};
struct Panic : std::logic_error { using std::logic_error::logic_error; };

inline void check_nul(const Bytes &b)
{
    if (b.find('\0') != Bytes::npos) throw Error(ErrorKind::InteriorNulByte, "nul byte found in provided data");
}

enum class SolutionWidth { Sat, Bit8, Bit16, Bit32, Bit64 };                  // src/prelude.rs:9-15
enum class InstructionSet { Best, SSE2, SSE41, AVX2, AltiVec, Neon };         // src/prelude.rs:18-25

// ------------------------------------------------------------------------------ Matrix --
class Matrix {                                                                // src/matrix/mod.rs:25-312
public:
    const parasail_matrix_t *inner = nullptr;
    bool builtin = false;

    static Matrix create(const Bytes &alphabet, int match_score, int mismatch_score)
    {
        if (!(match_score >= 0 && mismatch_score <= 0))
            throw Panic("Match score should be a positive integer and mismatch score should be a negative integer.");
        if (alphabet.empty()) throw Panic("Alphabet should not be empty.");
        check_nul(alphabet);
        return Matrix(parasail_matrix_create(alphabet.c_str(), match_score, mismatch_score), false);
    }
    static Matrix from(const std::string &name)
    {
        if (name.empty()) throw Panic("Matrix name should not be empty.");
        check_nul(name);
        const parasail_matrix_t *m = parasail_matrix_lookup(name.c_str());
        if (!m) throw Error(ErrorKind::FailedLookup, name);
        return Matrix(m, true);
    }
    static Matrix from_file(const std::string &file)
    {
        FILE *fh = fopen(file.c_str(), "r");
        if (!fh) throw Error(ErrorKind::FileNotFound, file);
        fclose(fh);
        parasail_matrix_t *m = parasail_matrix_from_file(file.c_str());
        if (!m) throw Error(ErrorKind::NullMatrix, file);
        return Matrix(m, false);
    }
    static Matrix create_pssm(const std::string &alphabet, const std::vector<int> &values, int rows)
    {
        check_nul(alphabet);
        std::vector<int> v(values);
        if (v.size() < alphabet.size() * (size_t)rows) v.resize(alphabet.size() * (size_t)rows, 0);
        parasail_matrix_t *m = parasail_matrix_pssm_create(alphabet.c_str(), v.data(), rows);
        if (!m) throw Error(ErrorKind::NullMatrix, "pssm_create");
        return Matrix(m, false);
    }
    Matrix to_pssm(const Bytes &pssm_query) const
    {
        if (pssm_query.empty()) throw Panic("PSSM query sequence should not be empty.");
        check_nul(pssm_query);
        if (inner->type != 0) throw Error(ErrorKind::NotSquare, "to_pssm");
        parasail_matrix_t *m = parasail_matrix_convert_square_to_pssm(inner, pssm_query.c_str(), (int)pssm_query.size());
        if (!m) throw Error(ErrorKind::NullMatrix, "to_pssm");
        return Matrix(m, false);
    }
    void set_value(int row, int col, int value)
    {
        if (builtin) throw Error(ErrorKind::NotBuiltIn, "set_value");
        const int size = inner->size - 2;
        if (size < 0) throw Error(ErrorKind::NullMatrix, "set_value");
        if (row < 0 || row > size || col < 0 || col > size) throw Error(ErrorKind::InvalidIndex, "set_value");
        parasail_matrix_set_value(const_cast<parasail_matrix_t *>(inner), row, col, value);
    }
    static Matrix default_() { return create("ACGTA", 1, -1); }               // src/matrix/mod.rs:246-250
    Matrix clone() const { return Matrix(parasail_matrix_copy(inner), false); }
    std::string to_string() const                                             // Display, :253-268
    {
        std::string s;
        for (int i = 0; i < inner->length; ++i) {
            for (int j = 0; j < inner->size; ++j) s += std::to_string(inner->matrix[i * inner->size + j]) + " ";
            s += "\n";
        }
        return s;
    }

    Matrix(Matrix &&o) noexcept : inner(o.inner), builtin(o.builtin) { o.inner = nullptr; }
    Matrix &operator=(Matrix &&o) noexcept { drop(); inner = o.inner; builtin = o.builtin; o.inner = nullptr; return *this; }
    Matrix(const Matrix &) = delete;
    Matrix &operator=(const Matrix &) = delete;
    ~Matrix() { drop(); }

private:
    Matrix(const parasail_matrix_t *m, bool b) : inner(m), builtin(b) {}
    void drop() { if (inner && !builtin) parasail_matrix_free(const_cast<parasail_matrix_t *>(inner)); inner = nullptr; }
};

// ----------------------------------------------------------------------------- Profile --
class Profile {                                                               // src/profile/mod.rs:281-395
public:
    parasail_profile_t *inner = nullptr;
    bool use_stats = false;
    int query_len = 0;

    static Profile new_(const Bytes &query, bool with_stats, const Matrix &matrix)
    {
        if (query.empty()) throw Error(ErrorKind::QueryIsEmpty, "Profile::new");
        check_nul(query);
        parasail_profile_t *p = with_stats ? parasail_profile_create_stats_sat(query.c_str(), (int)query.size(), matrix.inner)
                                           : parasail_profile_create_sat(query.c_str(), (int)query.size(), matrix.inner);
        if (!p) throw Error(ErrorKind::NullProfile, "Profile::new");
        return Profile(p, with_stats, (int)query.size());
    }
    static Profile new_ssw(const Bytes &query, const Matrix &matrix, int8_t score_size)
    {
        if (query.empty()) throw Panic("Query sequence has length 0.");
        check_nul(query);
        parasail_profile_t *p = parasail_ssw_init(query.c_str(), (int)query.size(), matrix.inner, score_size);
        if (!p) throw Error(ErrorKind::NullProfile, "Profile::new_ssw");
        return Profile(p, true, (int)query.size());
    }
    Profile() = default;                                                      // Default = null profile, :365-373
    bool is_null() const { return inner == nullptr; }
    Profile(Profile &&o) noexcept : inner(o.inner), use_stats(o.use_stats), query_len(o.query_len) { o.inner = nullptr; }
    Profile &operator=(Profile &&o) noexcept
    {
        if (inner) parasail_profile_free(inner);
        inner = o.inner; use_stats = o.use_stats; query_len = o.query_len; o.inner = nullptr; return *this;
    }
    Profile(const Profile &) = delete;
    ~Profile() { if (inner) parasail_profile_free(inner); }

private:
    Profile(parasail_profile_t *p, bool s, int n) : inner(p), use_stats(s), query_len(n) {}
    friend class ProfileBuilder;
};

class ProfileBuilder {                                                        // src/profile/mod.rs:42-278
public:
    ProfileBuilder(const Bytes &query, const Matrix &matrix) : query_(query), matrix_(matrix) {}
    ProfileBuilder &use_stats() { stats_ = true; return *this; }
    ProfileBuilder &solution_width(SolutionWidth w) { width_ = w; return *this; }
    ProfileBuilder &instruction_set(InstructionSet i) { isa_ = i; return *this; }
    Profile build() const
    {
        check_nul(query_);
        // On the GPU the ISA slot is meaningless: every ISA-suffixed creator is an alias (include/parasail_amd.h).
        parasail_pcreator_t *f = lookup();
        parasail_profile_t *p = f(query_.c_str(), (int)query_.size(), matrix_.inner);
        if (!p) throw Error(ErrorKind::NullProfile, "ProfileBuilder::build");
        return Profile(p, stats_, (int)query_.size());
    }

private:
    parasail_pcreator_t *lookup() const
    {
        (void)isa_;
        switch (width_) {
        case SolutionWidth::Sat: return stats_ ? parasail_profile_create_stats_sat : parasail_profile_create_sat;
        case SolutionWidth::Bit8: return stats_ ? parasail_profile_create_stats_8 : parasail_profile_create_8;
        case SolutionWidth::Bit16: return stats_ ? parasail_profile_create_stats_16 : parasail_profile_create_16;
        case SolutionWidth::Bit32: return stats_ ? parasail_profile_create_stats_32 : parasail_profile_create_32;
        default: return stats_ ? parasail_profile_create_stats_64 : parasail_profile_create_64;
        }
    }
    Bytes query_;
    const Matrix &matrix_;
    bool stats_ = false;
    SolutionWidth width_ = SolutionWidth::Sat;
    InstructionSet isa_ = InstructionSet::Best;
};

// ---------------------------------------------------------------------- tables / flags --
namespace TraceFlags {                                                        // src/alignment/table.rs:127-142
enum : int { ZERO_MASK = 120, E_MASK = 103, F_MASK = 31, ZERO = 0, INS = 1, DEL = 2, DIAG = 4,
             DIAG_E = 8, INS_E = 16, DIAG_F = 32, DEL_F = 64 };
}
template <typename T> class TableView {                                       // src/alignment/table.rs:33-108, :197-300
public:
    TableView(const T *d, size_t r, size_t c) : data_(d), rows_(r), cols_(c) {}
    bool get(size_t row, size_t col, int *out) const
    {
        if (row < rows_ && col < cols_) { *out = (int)data_[row * cols_ + col]; return true; }
        return false;
    }
    size_t rows() const { return rows_; }
    size_t cols() const { return cols_; }
    const T *as_slice() const { return data_; }
    size_t len() const { return rows_ * cols_; }
    int last() const { return (int)data_[rows_ * cols_ - 1]; }
private:
    const T *data_; size_t rows_, cols_;
};
using Table = TableView<int>;
using TracebackTable = TableView<int8_t>;
struct Traceback { std::string query, comparison, reference; };

// --------------------------------------------------------------------------- Alignment --
class Alignment {                                                             // src/alignment/mod.rs:54-504
public:
    // The reference keeps a raw matrix pointer with no lifetime tie (src/alignment/mod.rs:54-60), so a
    // result outliving its aligner dangles in get_cigar(); here the result co-owns the matrix instead.
    Alignment(parasail_result_t *r, std::shared_ptr<Matrix> m, int ql, int rl)
        : inner(r), matrix(m->inner), query_len(ql), ref_len(rl), keep_(std::move(m)) {}
    Alignment(Alignment &&o) noexcept
        : inner(o.inner), matrix(o.matrix), query_len(o.query_len), ref_len(o.ref_len), keep_(std::move(o.keep_)) { o.inner = nullptr; }
    Alignment(const Alignment &) = delete;
    ~Alignment() { if (inner) parasail_result_free(inner); }

    int get_score() const { return parasail_result_get_score(inner); }
    int get_end_query() const { return parasail_result_get_end_query(inner); }
    int get_end_ref() const { return parasail_result_get_end_ref(inner); }
    int get_matches() const { need(is_stats(), ErrorKind::NoStats, "get_matches()"); return parasail_result_get_matches(inner); }
    int get_similar() const { return parasail_result_get_similar(inner); }    // no guard in the reference, :87-89
    int get_length() const { need(is_stats(), ErrorKind::NoStats, "get_length()"); return parasail_result_get_length(inner); }

    Table get_score_table() const { need(is_table() || is_stats_table(), ErrorKind::NoTable, "get_score_table()"); return tab(parasail_result_get_score_table(inner)); }
    Table get_matches_table() const { need(is_stats_table(), ErrorKind::NoStatsTable, "get_matches_table()"); return tab(parasail_result_get_matches_table(inner)); }
    Table get_similar_table() const { need(is_stats_table(), ErrorKind::NoStatsTable, "get_similar_table()"); return tab(parasail_result_get_similar_table(inner)); }
    Table get_length_table() const { need(is_stats_table(), ErrorKind::NoStatsTable, "get_length_table()"); return tab(parasail_result_get_length_table(inner)); }
    std::vector<int> get_score_row() const { need(is_rowcol() || is_stats_rowcol(), ErrorKind::NoRowCol, "get_score_row()"); return vec(parasail_result_get_score_row(inner), ref_len); }
    std::vector<int> get_matches_row() const { need(is_stats_rowcol(), ErrorKind::NoRowCol, "get_matches_row()"); return vec(parasail_result_get_matches_row(inner), ref_len); }
    std::vector<int> get_similar_row() const { need(is_stats_rowcol(), ErrorKind::NoRowCol, "get_similar_row()"); return vec(parasail_result_get_similar_row(inner), ref_len); }
    std::vector<int> get_length_row() const { need(is_stats_rowcol(), ErrorKind::NoRowCol, "get_length_row()"); return vec(parasail_result_get_length_row(inner), ref_len); }
    std::vector<int> get_score_col() const { need(is_rowcol() || is_stats_rowcol(), ErrorKind::NoRowCol, "get_score_col()"); return vec(parasail_result_get_score_col(inner), query_len); }
    std::vector<int> get_matches_col() const { need(is_stats_rowcol(), ErrorKind::NoRowCol, "get_matches_col()"); return vec(parasail_result_get_matches_col(inner), query_len); }
    std::vector<int> get_similar_col() const { need(is_stats_rowcol(), ErrorKind::NoRowCol, "get_similar_col()"); return vec(parasail_result_get_similar_col(inner), query_len); }
    std::vector<int> get_length_col() const { need(is_stats_rowcol(), ErrorKind::NoRowCol, "get_length_col()"); return vec(parasail_result_get_length_col(inner), query_len); }
    TracebackTable get_trace_table() const
    {
        need(is_trace(), ErrorKind::NoTrace, "get_trace_table()");
        return TracebackTable(reinterpret_cast<const int8_t *>(parasail_result_get_trace_table(inner)), (size_t)query_len, (size_t)ref_len);
    }
    void print_traceback(const Bytes &query, const Bytes &reference) const
    {
        if (is_trace())
            parasail_traceback_generic(query.c_str(), (int)query.size(), reference.c_str(), (int)reference.size(),
                                       "Query:", "Target:", matrix, inner, '|', ' ', ' ', 80, 7, 1);
        else
            printf("Alignment string is not available without traceback enabled. Consider using the `use_trace` method on AlignerBuilder.\n");
    }
    Traceback get_traceback_strings(const Bytes &query, const Bytes &reference) const
    {
        need(is_trace(), ErrorKind::NoTrace, "get_traceback_strings()");
        check_nul(query); check_nul(reference);
        parasail_traceback_t *tb = parasail_result_get_traceback(inner, query.c_str(), (int)query.size(), reference.c_str(),
                                                                 (int)reference.size(), matrix, '|', ' ', ' ');
        if (!tb) throw Error(ErrorKind::NoTrace, "get_traceback_strings()");
        Traceback t{tb->query, tb->comp, tb->ref};
        parasail_traceback_free(tb);
        return t;
    }
    std::string get_cigar(const Bytes &query, const Bytes &reference) const
    {
        need(is_trace(), ErrorKind::NoTrace, "get_cigar()");
        check_nul(query); check_nul(reference);
        parasail_cigar_t *c = parasail_result_get_cigar(inner, query.c_str(), (int)query.size(), reference.c_str(),
                                                       (int)reference.size(), matrix);
        if (!c) throw Error(ErrorKind::NoTrace, "get_cigar()");
        char *s = parasail_cigar_decode(c);
        std::string out(s ? s : "");
        free(s);
        parasail_cigar_free(c);
        return out;
    }
    bool is_global() const { return parasail_result_is_nw(inner) != 0; }
    bool is_semi_global() const { return parasail_result_is_sg(inner) != 0; }
    bool is_local() const { return parasail_result_is_sw(inner) != 0; }
    bool is_saturated() const { return parasail_result_is_saturated(inner) != 0; }
    bool is_banded() const { return parasail_result_is_banded(inner) != 0; }
    bool is_scan() const { return parasail_result_is_scan(inner) != 0; }
    bool is_striped() const { return parasail_result_is_striped(inner) != 0; }
    bool is_diag() const { return parasail_result_is_diag(inner) != 0; }
    bool is_blocked() const { return parasail_result_is_blocked(inner) != 0; }
    bool is_stats() const { return parasail_result_is_stats(inner) != 0; }
    bool is_stats_table() const { return parasail_result_is_stats_table(inner) != 0; }
    bool is_table() const { return parasail_result_is_table(inner) != 0; }
    bool is_rowcol() const { return parasail_result_is_rowcol(inner) != 0; }
    bool is_stats_rowcol() const { return parasail_result_is_stats_rowcol(inner) != 0; }
    bool is_trace() const { return parasail_result_is_trace(inner) != 0; }

    parasail_result_t *inner;
    const parasail_matrix_t *matrix;
    int query_len, ref_len;

private:
    static void need(bool ok, ErrorKind k, const char *fn) { if (!ok) throw Error(k, fn); }
    Table tab(const int *p) const { return Table(p, (size_t)query_len, (size_t)ref_len); }
    static std::vector<int> vec(const int *p, int n) { return std::vector<int>(p, p + n); }
    std::shared_ptr<Matrix> keep_;
};

class SSWResult {                                                             // src/alignment/mod.rs:506-551
public:
    explicit SSWResult(parasail_result_ssw_t *r) : inner(r) {}
    SSWResult(SSWResult &&o) noexcept : inner(o.inner) { o.inner = nullptr; }
    SSWResult(const SSWResult &) = delete;
    ~SSWResult() { if (inner) parasail_result_ssw_free(inner); }
    uint16_t score() const { return inner->score1; }
    int ref_start() const { return inner->ref_begin1; }
    int ref_end() const { return inner->ref_end1; }
    int query_start() const { return inner->read_begin1; }
    int query_end() const { return inner->read_end1; }
    uint32_t *cigar() const { return inner->cigar; }
    int cigar_len() const { return inner->cigarLen; }
    parasail_result_ssw_t *inner;
};

// ----------------------------------------------------------------------------- Aligner --
class Aligner;
class AlignerBuilder {                                                        // src/aligner/mod.rs:67-370
public:
    AlignerBuilder() : matrix_(std::make_shared<Matrix>(Matrix::default_())), profile_(std::make_shared<Profile>()) {}
    AlignerBuilder &global() { mode_ = "nw"; return *this; }
    AlignerBuilder &semi_global() { mode_ = "sg"; return *this; }
    AlignerBuilder &local() { mode_ = "sw"; return *this; }
    AlignerBuilder &solution_width(int w) { width_ = std::to_string(w); return *this; }
    AlignerBuilder &matrix(Matrix m) { matrix_ = std::make_shared<Matrix>(std::move(m)); return *this; }
    AlignerBuilder &gap_open(int v) { gap_open_ = v; return *this; }
    AlignerBuilder &gap_extend(int v) { gap_extend_ = v; return *this; }
    AlignerBuilder &profile(Profile p) { profile_ = std::make_shared<Profile>(std::move(p)); return *this; }
    AlignerBuilder &allow_query_gaps(std::vector<std::string> g) { qgaps_ = std::move(g); return *this; }
    AlignerBuilder &allow_ref_gaps(std::vector<std::string> g) { rgaps_ = std::move(g); return *this; }
    AlignerBuilder &striped() { vec_ = "_striped"; return *this; }
    AlignerBuilder &scan() { vec_ = "_scan"; return *this; }
    AlignerBuilder &diag() { vec_ = "_diag"; return *this; }
    AlignerBuilder &use_stats() { stats_ = "_stats"; trace_.clear(); return *this; }            // :213-223
    AlignerBuilder &use_table() { table_ = "_table"; trace_.clear(); return *this; }            // :228-237
    AlignerBuilder &use_last_rowcol() { table_ = "_rowcol"; return *this; }                     // :243-246
    AlignerBuilder &use_trace() { trace_ = "_trace"; table_.clear(); stats_.clear(); return *this; }   // :251-267
    AlignerBuilder &bandwidth(int k) { has_band_ = true; band_ = k; return *this; }

    std::string get_parasail_fn_name() const                                  // :289-331
    {
        std::string sg;
        if (mode_ == "sg") {
            sg = allowed("q", qgaps_) + allowed("d", rgaps_);
            if (sg == "_qx_dx") sg.clear();
        }
        std::string prof, stats;
        if (profile_->is_null()) stats = stats_;
        else {
            if (!(vec_ == "_striped" || vec_ == "_scan"))
                throw Panic("Vectorization strategy must be striped or scan for alignment with a profile.");
            prof = "_profile";
            stats = profile_->use_stats ? "_stats" : "";
        }
        return mode_ + sg + trace_ + stats + table_ + vec_ + prof + "_" + width_;
    }
    Aligner build() const;

private:
    static bool has(const std::vector<std::string> &v, const char *s)
    {
        for (auto &x : v) if (x == s) return true;
        return false;
    }
    static std::string allowed(const char *prefix, const std::vector<std::string> &g)   // :270-286
    {
        if (g.empty()) return "";
        if (has(g, "prefix") && has(g, "suffix")) return std::string("_") + prefix + "x";
        if (has(g, "prefix")) return std::string("_") + prefix + "b";
        if (has(g, "suffix")) return std::string("_") + prefix + "e";
        return "";
    }
    std::string mode_ = "nw", width_ = "sat", vec_ = "_striped", stats_, table_, trace_;
    std::shared_ptr<Matrix> matrix_;
    int gap_open_ = 0, gap_extend_ = 0;       // defaults are 0/0 (:92-93), whatever the doc comments say
    std::shared_ptr<Profile> profile_;
    std::vector<std::string> qgaps_, rgaps_;
    bool has_band_ = false; int band_ = 0;
    friend class Aligner;
};

class Aligner {                                                               // src/aligner/mod.rs:372-535
public:
    static AlignerBuilder builder() { return AlignerBuilder(); }              // Rust: Aligner::new()

    Alignment align(const Bytes *query, const Bytes &reference) const         // :397-452
    {
        check_nul(reference);
        if (fn_) {
            if (!query) throw Panic("Query sequence is required for alignment without a profile.");
            check_nul(*query);
            parasail_result_t *r = fn_(query->c_str(), (int)query->size(), reference.c_str(), (int)reference.size(),
                                       gap_open, gap_extend, matrix->inner);
            return Alignment(r, matrix, (int)query->size(), (int)reference.size());
        }
        parasail_result_t *r = pfn_(profile_->inner, reference.c_str(), (int)reference.size(), gap_open, gap_extend);
        return Alignment(r, matrix, profile_->query_len, (int)reference.size());
    }
    Alignment banded_nw(const Bytes &query, const Bytes &reference) const     // :457-489
    {
        check_nul(reference); check_nul(query);
        if (!has_band_) throw Error(ErrorKind::NoBandwidth, "banded_nw");
        parasail_result_t *r = parasail_nw_banded(query.c_str(), (int)query.size(), reference.c_str(), (int)reference.size(),
                                                  gap_open, gap_extend, band_, matrix->inner);
        return Alignment(r, matrix, (int)query.size(), (int)reference.size());
    }
    SSWResult ssw(const Bytes *query, const Bytes &reference) const           // :492-529
    {
        check_nul(reference);
        if (!query) throw Panic("Query sequence is required for SSW alignment for now.");
        check_nul(*query);
        return SSWResult(parasail_ssw(query->c_str(), (int)query->size(), reference.c_str(), (int)reference.size(),
                                      gap_open, gap_extend, matrix->inner));
    }
    // additive: many independent pairs per call (include/parasail_amd.h, pmx_align_batch)
    std::vector<pmx_record_t> align_batch(const std::vector<Bytes> &queries, const std::vector<Bytes> &refs,
                                          std::vector<pmx_stats_t> *stats = nullptr) const
    {
        if (queries.size() != refs.size()) throw Error(ErrorKind::Batch, "queries and references differ in count");
        std::string qb, rb; std::vector<int64_t> qo(1, 0), ro(1, 0);
        for (auto &q : queries) { qb += q; qo.push_back((int64_t)qb.size()); }
        for (auto &r : refs) { rb += r; ro.push_back((int64_t)rb.size()); }
        pmx_config_t cfg = config_;
        cfg.matrix = matrix->inner;
        std::vector<pmx_record_t> out(refs.size());
        if (cfg.want & PMX_WANT_STATS) { if (!stats) throw Error(ErrorKind::Batch, "stats aligner needs a stats vector"); stats->resize(refs.size()); }
        const int rc = pmx_align_batch(&cfg, (int64_t)refs.size(), (const uint8_t *)qb.data(), qo.data(), (const uint8_t *)rb.data(),
                                       ro.data(), out.data(), (cfg.want & PMX_WANT_STATS) ? stats->data() : nullptr);
        if (rc) throw Error(ErrorKind::Batch, pmx_last_error());
        return out;
    }
    // additive: the profile arm for many references (pmx_align_profile_batch; the aligner was built with .profile())
    std::vector<pmx_record_t> align_profile_batch(const std::vector<Bytes> &refs, std::vector<pmx_stats_t> *stats = nullptr) const
    {
        if (!profile_) throw Error(ErrorKind::Batch, "aligner has no profile");
        std::string rb; std::vector<int64_t> ro(1, 0);
        for (auto &r : refs) { rb += r; ro.push_back((int64_t)rb.size()); }
        pmx_config_t cfg = config_;
        cfg.matrix = matrix->inner;
        std::vector<pmx_record_t> out(refs.size());
        if (cfg.want & PMX_WANT_STATS) { if (!stats) throw Error(ErrorKind::Batch, "stats aligner needs a stats vector"); stats->resize(refs.size()); }
        const int rc = pmx_align_profile_batch(&cfg, profile_->inner, (int64_t)refs.size(), (const uint8_t *)rb.data(), ro.data(),
                                               out.data(), (cfg.want & PMX_WANT_STATS) ? stats->data() : nullptr);
        if (rc) throw Error(ErrorKind::Batch, pmx_last_error());
        return out;
    }
    // additive: records + CIGAR text per pair, traceback done on the device (pmx_align_batch_cigar)
    std::vector<pmx_record_t> align_batch_cigar(const std::vector<Bytes> &queries, const std::vector<Bytes> &refs,
                                                std::vector<std::string> &cigars) const
    {
        if (queries.size() != refs.size()) throw Error(ErrorKind::Batch, "queries and references differ in count");
        std::string qb, rb; std::vector<int64_t> qo(1, 0), ro(1, 0);
        for (auto &q : queries) { qb += q; qo.push_back((int64_t)qb.size()); }
        for (auto &r : refs) { rb += r; ro.push_back((int64_t)rb.size()); }
        pmx_config_t cfg = config_;
        cfg.matrix = matrix->inner;
        cfg.want &= ~PMX_WANT_STATS;
        std::vector<pmx_record_t> out(refs.size());
        std::vector<int64_t> coff(refs.size() + 1);
        char *text = nullptr;
        const int rc = pmx_align_batch_cigar(&cfg, (int64_t)refs.size(), (const uint8_t *)qb.data(), qo.data(), (const uint8_t *)rb.data(),
                                             ro.data(), out.data(), &text, coff.data());
        if (rc) throw Error(ErrorKind::Batch, pmx_last_error());
        cigars.resize(refs.size());
        for (size_t k = 0; k < refs.size(); ++k) cigars[k].assign(text + coff[k], text + coff[k + 1]);
        pmx_free(text);
        return out;
    }

    std::shared_ptr<Matrix> matrix;
    int gap_open = 0, gap_extend = 0;
    std::string vec_strategy;

private:
    Aligner() = default;
    parasail_function_t *fn_ = nullptr;
    parasail_pfunction_t *pfn_ = nullptr;
    std::shared_ptr<Profile> profile_;
    bool has_band_ = false; int band_ = 0;
    pmx_config_t config_{};
    friend class AlignerBuilder;
};

inline Aligner AlignerBuilder::build() const                                  // :339-369
{
    const std::string name = get_parasail_fn_name();
    Aligner a;
    if (profile_->is_null()) a.fn_ = parasail_lookup_function(name.c_str());
    else a.pfn_ = parasail_lookup_pfunction(name.c_str());
    if (!a.fn_ && !a.pfn_) throw Panic("Parasail function: " + name + ", not found.");
    a.matrix = matrix_; a.gap_open = gap_open_; a.gap_extend = gap_extend_; a.profile_ = profile_;
    a.vec_strategy = vec_; a.has_band_ = has_band_; a.band_ = band_;
    pmx_config_t c{};
    c.mode = mode_ == "nw" ? PMX_MODE_NW : mode_ == "sg" ? PMX_MODE_SG : PMX_MODE_SW;
    if (mode_ == "sg") {
        const std::string q = allowed("q", qgaps_), d = allowed("d", rgaps_);
        if (q.empty() && d.empty()) c.sg_flags = PMX_SG_ALL;
        else {
            if (q == "_qb" || q == "_qx") c.sg_flags |= PMX_SG_QB;
            if (q == "_qe" || q == "_qx") c.sg_flags |= PMX_SG_QE;
            if (d == "_db" || d == "_dx") c.sg_flags |= PMX_SG_DB;
            if (d == "_de" || d == "_dx") c.sg_flags |= PMX_SG_DE;
        }
    }
    c.open = gap_open_; c.extend = gap_extend_;
    c.width = width_ == "sat" ? 0 : atoi(width_.c_str());
    c.want = (profile_->is_null() ? !stats_.empty() : profile_->use_stats) ? PMX_WANT_STATS : 0;
    a.config_ = c;
    return a;
}

}  // namespace parasail_rs
