// pmx_api.hip -- the C ABI of libparasail_amd.so (declared in include/parasail_amd.h).
//
// Host side of the drop-in boundary: the `parasail_*` symbols parasail-rs binds through
// libparasail-sys (/root/reference/src/aligner/mod.rs:4-7, src/alignment/mod.rs:6-23,
// src/matrix/mod.rs:7-11, src/profile/mod.rs:5-32) plus the additive `pmx_*` batch entries.
// All DP arithmetic is done by the HIP kernels (pmx_sw16.hip, pmx_general.hip); this file
// owns handles, dispatch-name parsing, device staging and result marshalling only.
#include "pmx_common.h"
#include "pmx_switches.h"
#include <chrono>
#include <future>
#include "pmx_matrices.h"

#include <cctype>
#include <dlfcn.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

// ---- the switch table (pmx_switches.h) ------------------------------------------------------------------------------
static const PmxSwitchDoc g_switches[] = {
#define X(name, kind, what) {name, kind, what},
    PMX_SWITCH_TABLE(X)
#undef X
};
const char *pmx_env(const char *name)
{
    for (const PmxSwitchDoc &d : g_switches)
        if (!strcmp(d.name, name)) return getenv(name);
    fprintf(stderr, "libparasail_amd: environment switch %s is not in pmx_switches.h\n", name);
    abort();
}
// "NAME\tkind\twhat\n" per switch (static storage)
extern "C" const char *pmx_switches(void)
{
    static const std::string text = []() {
        std::string t;
        for (const PmxSwitchDoc &d : g_switches) { t += d.name; t += '\t'; t += d.kind; t += '\t'; t += d.what; t += '\n'; }
        return t;
    }();
    return text.c_str();
}

// ============================================================================ errors ====
static thread_local char g_err[512] = "";
static void set_err(const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}
extern "C" const char *pmx_last_error(void) { return g_err; }
extern "C" const char *pmx_version(void) { return "parasail_amd 0.1.0 (gfx950)"; }

// The product path has no CPU fallback: a failing HIP call in a function whose signature
// cannot report errors (the reference never checks alignment results for NULL,
// src/aligner/mod.rs:411-429) is fatal and loud.
[[noreturn]] static void die(const char *what, hipError_t e)
{
    fprintf(stderr, "libparasail_amd: fatal: %s: %s (no CPU fallback exists)\n", what,
            e == hipSuccess ? "" : hipGetErrorString(e));
    abort();
}
#define HIP_OR_DIE(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) die(#expr, e__); } while (0)
#define HIP_OR_RET(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { \
    set_err("%s: %s", #expr, hipGetErrorString(e__)); return -(int)e__; } } while (0)

extern "C" int pmx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" int pmx_set_device(int device)
{
    HIP_OR_RET(hipSetDevice(device));
    return 0;
}

// =========================================================================== matrices ===
struct MatrixBox {               // owns everything a non-builtin parasail_matrix_t points to
    parasail_matrix_t m;
    std::vector<int> scores, mapper;
    std::string name, alphabet, query;
};

static void finish_box(MatrixBox *b, int type, int length, int size, bool user)
{
    b->m.name = b->name.c_str();
    b->m.matrix = b->scores.data();
    b->m.mapper = b->mapper.data();
    b->m.size = size;
    int mx = INT32_MIN, mn = INT32_MAX;
    for (int v : b->scores) { mx = v > mx ? v : mx; mn = v < mn ? v : mn; }
    b->m.max = mx; b->m.min = mn;
    b->m.user_matrix = user ? b->scores.data() : nullptr;
    b->m.type = type;
    b->m.length = length;
    b->m.alphabet = b->alphabet.c_str();
    b->m.query = b->query.empty() ? nullptr : b->query.c_str();
}

static std::mutex g_mx_mutex;
static std::unordered_map<const parasail_matrix_t *, MatrixBox *> g_boxes;   // live non-builtin matrices

static parasail_matrix_t *publish(MatrixBox *b)
{
    std::lock_guard<std::mutex> lk(g_mx_mutex);
    g_boxes[&b->m] = b;
    return &b->m;
}

static void fill_mapper(std::vector<int> &mapper, const std::string &alphabet, int wildcard)
{
    mapper.assign(256, wildcard);
    for (size_t i = 0; i < alphabet.size(); ++i) {
        const unsigned char c = (unsigned char)alphabet[i];
        mapper[(unsigned char)toupper(c)] = (int)i;
        mapper[(unsigned char)tolower(c)] = (int)i;
    }
}

// src/matrix/mod.rs:34-44.  size = alphabet + 1 wildcard row/col (bound size-2 at :228-236);
// wildcard scores 0; a repeated letter ("ACGTA", :248) maps to its last position.
extern "C" parasail_matrix_t *parasail_matrix_create(const char *alphabet, const int match, const int mismatch)
{
    if (!alphabet || !*alphabet) return nullptr;
    MatrixBox *b = new MatrixBox;
    b->alphabet = std::string(alphabet) + "*";
    b->name = "";
    const int n = (int)strlen(alphabet), size = n + 1;
    b->scores.assign((size_t)size * size, 0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) b->scores[(size_t)i * size + j] = (i == j) ? match : mismatch;
    fill_mapper(b->mapper, alphabet, n);
    finish_box(b, PARASAIL_MATRIX_TYPE_SQUARE, size, size, true);
    return publish(b);
}

static parasail_matrix_t g_blosum62, g_nuc44;
static std::vector<int> g_blosum62_mapper, g_nuc44_mapper;
static std::once_flag g_builtin_once;
static void init_builtins()
{
    fill_mapper(g_blosum62_mapper, std::string(pmx_blosum62_alphabet, 23), 23);
    g_blosum62_mapper[(unsigned char)'*'] = 23;
    g_blosum62.name = "blosum62";
    g_blosum62.matrix = pmx_blosum62_scores;
    g_blosum62.mapper = g_blosum62_mapper.data();
    g_blosum62.size = 24; g_blosum62.max = 11; g_blosum62.min = -4;
    g_blosum62.user_matrix = nullptr; g_blosum62.type = PARASAIL_MATRIX_TYPE_SQUARE;
    g_blosum62.length = 24; g_blosum62.alphabet = pmx_blosum62_alphabet; g_blosum62.query = nullptr;
    fill_mapper(g_nuc44_mapper, std::string(pmx_nuc44_alphabet, 15), 15);
    g_nuc44_mapper[(unsigned char)'*'] = 15;
    g_nuc44.name = "nuc44"; g_nuc44.matrix = pmx_nuc44_scores; g_nuc44.mapper = g_nuc44_mapper.data();
    g_nuc44.size = 16; g_nuc44.max = 5; g_nuc44.min = -5; g_nuc44.user_matrix = nullptr;
    g_nuc44.type = PARASAIL_MATRIX_TYPE_SQUARE; g_nuc44.length = 16; g_nuc44.alphabet = pmx_nuc44_alphabet; g_nuc44.query = nullptr;
}

extern "C" parasail_matrix_t *parasail_matrix_from_file(const char *filename);
extern "C" void parasail_matrix_free(parasail_matrix_t *matrix);

// The reference documents 66 built-in names (blosum{30..100}, pam{10..500}: src/matrix/mod.rs:46-50); the tables themselves live in
// the parasail C library, which is not in this image, and typing 64 of them from memory cannot be checked here.  Embedded:
// blosum62, nuc44 (both verified against files: tests/golden/blosum62.txt, the reference's own tests/square.txt).  Every other
// name resolves from $PMX_MATRIX_DIR/<name>[.txt|.mat] (NCBI format, the loader of parasail_matrix_from_file) and then
// behaves like a built-in: cached for the life of the process, never freed, set_value rejected.
//
// ONE table of the documented names with what their files must look like: the 20 residues + B Z X * of the NCBI files
// (24 symbols; a file with further columns, e.g. J / U / O of newer NCBI releases, is accepted as long as these are there).
static const char *const pmx_documented_residues = "ARNDCQEGHILKMFPSTWYV";
static bool documented_matrix_name(const std::string &lname, std::string *family)
{
    static const int blosum[] = {30, 35, 40, 45, 50, 55, 60, 62, 65, 70, 75, 80, 85, 90, 95, 100};
    auto number = [&](const char *prefix, int *out) {
        const size_t pl = strlen(prefix);
        if (lname.compare(0, pl, prefix) != 0 || lname.size() == pl || lname.size() > pl + 3) return false;
        int v = 0;
        for (size_t i = pl; i < lname.size(); ++i) { if (!isdigit((unsigned char)lname[i])) return false; v = 10 * v + (lname[i] - '0'); }
        if (lname[pl] == '0') return false;
        *out = v; return true;
    };
    int v = 0;
    if (number("blosum", &v)) { for (int b : blosum) if (b == v) { if (family) *family = "blosum"; return true; } return false; }
    if (number("pam", &v)) { if (v >= 10 && v <= 500 && v % 10 == 0) { if (family) *family = "pam"; return true; } return false; }
    return false;
}
// the 66 names, for tests and for `pmx_last_error()` ("known name, file missing" vs "unknown name")
extern "C" int pmx_documented_matrix_names(char *buf, int cap)
{
    std::string all;
    static const int blosum[] = {30, 35, 40, 45, 50, 55, 60, 62, 65, 70, 75, 80, 85, 90, 95, 100};
    for (int b : blosum) all += "blosum" + std::to_string(b) + "\n";
    for (int p = 10; p <= 500; p += 10) all += "pam" + std::to_string(p) + "\n";
    if (buf && cap > 0) { strncpy(buf, all.c_str(), (size_t)cap - 1); buf[cap - 1] = 0; }
    return (int)all.size() + 1;
}

// Load-time self-check of a file that claims a documented name: a damaged or mislabelled table is refused (a silently wrong table
// is worse than a failed lookup).  Square; the 20 residues, B, Z, X and * present; symmetric; positive diagonal on the residues;
// every `*` score against a letter is the table's minimum; B lies between N and D, Z between Q and E, X inside the residues' range.
static bool documented_matrix_selfcheck(const parasail_matrix_t *m, std::string *why)
{
    if (m->type != PARASAIL_MATRIX_TYPE_SQUARE) { *why = "not a square matrix"; return false; }
    const int n = m->size;
    auto idx = [&](char c) -> int { const char *p = m->alphabet ? strchr(m->alphabet, c) : nullptr; return p ? (int)(p - m->alphabet) : -1; };
    auto at = [&](int a, int b) { return m->matrix[(size_t)a * n + b]; };
    int res[20];
    for (int r = 0; r < 20; ++r) { res[r] = idx(pmx_documented_residues[r]); if (res[r] < 0) { *why = std::string("residue ") + pmx_documented_residues[r] + " missing"; return false; } }
    const int iB = idx('B'), iZ = idx('Z'), iX = idx('X'), iS = idx('*');
    if (iB < 0 || iZ < 0 || iX < 0 || iS < 0) { *why = "one of B Z X * missing"; return false; }
    for (int a = 0; a < n; ++a) for (int b = 0; b < a; ++b) if (at(a, b) != at(b, a)) {
        *why = std::string("not symmetric at ") + m->alphabet[a] + "/" + m->alphabet[b]; return false; }
    for (int r = 0; r < 20; ++r) if (at(res[r], res[r]) <= 0) { *why = std::string("diagonal of ") + pmx_documented_residues[r] + " is not positive"; return false; }
    for (int a = 0; a < n; ++a) if (a != iS && at(iS, a) != m->min) { *why = std::string("* against ") + m->alphabet[a] + " is not the table's minimum"; return false; }
    auto between = [&](int amb, char c1, char c2, const char *nm) {
        const int i1 = idx(c1), i2 = idx(c2);
        for (int r = 0; r < 20; ++r) {
            const int lo = std::min(at(i1, res[r]), at(i2, res[r])), hi = std::max(at(i1, res[r]), at(i2, res[r]));
            if (at(amb, res[r]) < lo || at(amb, res[r]) > hi) { *why = std::string(nm) + " against " + pmx_documented_residues[r] + " is outside the range of the residues it stands for"; return false; }
        }
        return true;
    };
    if (!between(iB, 'N', 'D', "B") || !between(iZ, 'Q', 'E', "Z")) return false;
    for (int r = 0; r < 20; ++r) {
        int lo = INT32_MAX, hi = INT32_MIN;
        for (int q = 0; q < 20; ++q) { lo = std::min(lo, at(res[q], res[r])); hi = std::max(hi, at(res[q], res[r])); }
        if (at(iX, res[r]) < lo || at(iX, res[r]) > hi) { *why = std::string("X against ") + pmx_documented_residues[r] + " is outside the residues' range"; return false; }
    }
    return true;
}

static const parasail_matrix_t *lookup_in_matrix_dir(const std::string &lname)
{
    static std::mutex mx;
    static std::unordered_map<std::string, const parasail_matrix_t *> cache;
    for (char c : lname) if (!(isalnum((unsigned char)c) || c == '_' || c == '-' || c == '.')) { set_err("matrix lookup: '%s' is not a name", lname.c_str()); return nullptr; }
    if (lname.empty() || lname[0] == '.') { set_err("matrix lookup: empty name"); return nullptr; }
    const bool documented = documented_matrix_name(lname, nullptr);
    std::lock_guard<std::mutex> lk(mx);
    // $PMX_MATRIX_DIR, else the directory shipped beside the library: <libdir>/../matrices (parasail-rs_amd/matrices in this tree)
    std::string dir;
    const char *env = pmx_env("PMX_MATRIX_DIR");
    if (env && *env) dir = env;
    else {
        Dl_info info;
        if (dladdr((const void *)&lookup_in_matrix_dir, &info) && info.dli_fname) {
            const std::string so(info.dli_fname);
            const size_t slash = so.rfind('/');
            dir = (slash == std::string::npos ? std::string(".") : so.substr(0, slash)) + "/../matrices";
        }
    }
    if (dir.empty()) { set_err("matrix lookup: no matrix directory"); return nullptr; }
    const std::string key = dir + "\n" + lname;               // hits are remembered per directory; a miss is looked up again (a file may have arrived)
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    parasail_matrix_t *m = nullptr;
    for (const char *suffix : {"", ".txt", ".mat"}) {
        const std::string path = dir + "/" + lname + suffix;
        m = parasail_matrix_from_file(path.c_str());
        if (m) break;
    }
    if (!m) {
        if (documented) set_err("matrix lookup: '%s' is a documented name (src/matrix/mod.rs:46-50) whose table is not embedded, and no file %s/%s[.txt|.mat] "
                                "was found: put the NCBI file there (or set PMX_MATRIX_DIR)", lname.c_str(), dir.c_str(), lname.c_str());
        else set_err("matrix lookup: unknown matrix name '%s' (not embedded, not documented, no file in %s)", lname.c_str(), dir.c_str());
        return nullptr;
    }
    if (documented) {
        std::string why;
        if (!documented_matrix_selfcheck(m, &why)) {
            set_err("matrix lookup: the file found for '%s' in %s is refused: %s", lname.c_str(), dir.c_str(), why.c_str());
            parasail_matrix_free(m);
            return nullptr;
        }
    }
    MatrixBox *b = nullptr;
    {   // from here on it is a built-in: out of the table of caller-owned matrices, not writable
        std::lock_guard<std::mutex> lk2(g_mx_mutex);
        auto bi = g_boxes.find(m);
        if (bi != g_boxes.end()) { b = bi->second; g_boxes.erase(bi); }
    }
    if (b) { b->name = lname; b->m.name = b->name.c_str(); }
    m->user_matrix = nullptr;
    cache[key] = m;
    return m;
}

// src/matrix/mod.rs:57-73: NULL -> Error::FailedLookup.  Built-ins are static, never freed.
extern "C" const parasail_matrix_t *parasail_matrix_lookup(const char *matrixname)
{
    if (!matrixname) return nullptr;
    std::call_once(g_builtin_once, init_builtins);
    std::string s(matrixname);
    for (auto &c : s) c = (char)tolower((unsigned char)c);
    if (s == "blosum62") return &g_blosum62;
    if (s == "nuc44" || s == "dnafull") return &g_nuc44;      // EDNAFULL is the NUC.4.4 table under EMBOSS's name
    return lookup_in_matrix_dir(s);
}

// File formats: tests/square.txt:1-27 (square, trailing wildcard row/col) and
// tests/pssm.txt:1-17 (PSSM, optional leading residue column).
extern "C" parasail_matrix_t *parasail_matrix_from_file(const char *filename)
{
    if (!filename) return nullptr;
    FILE *fh = fopen(filename, "r");
    if (!fh) return nullptr;
    std::vector<std::string> header;
    std::vector<std::vector<std::string>> rows;
    char line[8192];
    while (fgets(line, sizeof line, fh)) {
        char *p = line;
        while (*p && isspace((unsigned char)*p)) ++p;
        if (!*p || *p == '#') continue;
        std::vector<std::string> tok;
        char *save = nullptr;
        for (char *t = strtok_r(p, " \t\r\n", &save); t; t = strtok_r(nullptr, " \t\r\n", &save)) tok.emplace_back(t);
        if (tok.empty()) continue;
        if (header.empty()) header = tok; else rows.push_back(tok);
    }
    fclose(fh);
    const int n = (int)header.size();
    if (n < 2 || rows.empty()) return nullptr;
    for (auto &h : header) if (h.size() != 1) return nullptr;
    bool square = (int)rows.size() == n;
    if (square) for (int i = 0; i < n; ++i)
        if ((int)rows[i].size() != n + 1 || rows[i][0] != header[i]) { square = false; break; }
    MatrixBox *b = new MatrixBox;
    b->name = filename;
    for (auto &h : header) b->alphabet += h;
    auto parse = [](const std::string &s, int *out) {
        char *end = nullptr; long v = strtol(s.c_str(), &end, 10);
        if (!end || *end) return false; *out = (int)v; return true;
    };
    if (square) {
        b->scores.resize((size_t)n * n);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j)
            if (!parse(rows[i][j + 1], &b->scores[(size_t)i * n + j])) { delete b; return nullptr; }
        fill_mapper(b->mapper, b->alphabet.substr(0, n - 1), n - 1);
        b->mapper[(unsigned char)b->alphabet[n - 1]] = n - 1;
        finish_box(b, PARASAIL_MATRIX_TYPE_SQUARE, n, n, true);
    } else {
        const int len = (int)rows.size(), size = n + 1;
        b->scores.assign((size_t)len * size, 0);
        for (int i = 0; i < len; ++i) {
            const int skip = (int)rows[i].size() == n + 1 ? 1 : 0;
            if ((int)rows[i].size() != n + skip) { delete b; return nullptr; }
            int mn = INT32_MAX;
            for (int j = 0; j < n; ++j) {
                int v; if (!parse(rows[i][j + skip], &v)) { delete b; return nullptr; }
                b->scores[(size_t)i * size + j] = v; mn = v < mn ? v : mn;
            }
            b->scores[(size_t)i * size + n] = mn;
            if (skip) b->query += rows[i][0];
        }
        b->alphabet += "*";
        fill_mapper(b->mapper, b->alphabet.substr(0, n), n);
        finish_box(b, PARASAIL_MATRIX_TYPE_PSSM, len, size, true);
    }
    return publish(b);
}

// src/matrix/mod.rs:154-169: `values` holds length * strlen(alphabet) scores, row-major.
extern "C" parasail_matrix_t *parasail_matrix_pssm_create(const char *alphabet, const int *values, const int length)
{
    if (!alphabet || !*alphabet || !values || length <= 0) return nullptr;
    const int n = (int)strlen(alphabet), size = n + 1;
    MatrixBox *b = new MatrixBox;
    b->alphabet = std::string(alphabet) + "*";
    b->scores.assign((size_t)length * size, 0);
    for (int i = 0; i < length; ++i) {
        int mn = INT32_MAX;
        for (int j = 0; j < n; ++j) { const int v = values[(size_t)i * n + j]; b->scores[(size_t)i * size + j] = v; mn = v < mn ? v : mn; }
        b->scores[(size_t)i * size + n] = mn;
    }
    fill_mapper(b->mapper, alphabet, n);
    finish_box(b, PARASAIL_MATRIX_TYPE_PSSM, length, size, true);
    return publish(b);
}

// src/matrix/mod.rs:180-212
extern "C" parasail_matrix_t *parasail_matrix_convert_square_to_pssm(const parasail_matrix_t *matrix,
                                                                   const char *s1, int s1Len)
{
    if (!matrix || matrix->type != PARASAIL_MATRIX_TYPE_SQUARE || !s1 || s1Len <= 0) return nullptr;
    MatrixBox *b = new MatrixBox;
    b->name = matrix->name ? matrix->name : "";
    b->alphabet = matrix->alphabet ? matrix->alphabet : "";
    b->query.assign(s1, (size_t)s1Len);
    const int size = matrix->size;
    b->scores.resize((size_t)s1Len * size);
    for (int i = 0; i < s1Len; ++i)
        memcpy(&b->scores[(size_t)i * size], &matrix->matrix[(size_t)matrix->mapper[(unsigned char)s1[i]] * size],
               sizeof(int) * (size_t)size);
    b->mapper.assign(matrix->mapper, matrix->mapper + 256);
    finish_box(b, PARASAIL_MATRIX_TYPE_PSSM, s1Len, size, true);
    return publish(b);
}

// src/matrix/mod.rs:279-294
extern "C" parasail_matrix_t *parasail_matrix_copy(const parasail_matrix_t *matrix)
{
    if (!matrix) return nullptr;
    MatrixBox *b = new MatrixBox;
    b->name = matrix->name ? matrix->name : "";
    b->alphabet = matrix->alphabet ? matrix->alphabet : "";
    if (matrix->query) b->query = matrix->query;
    b->scores.assign(matrix->matrix, matrix->matrix + (size_t)matrix->length * matrix->size);
    b->mapper.assign(matrix->mapper, matrix->mapper + 256);
    finish_box(b, matrix->type, matrix->length, matrix->size, true);
    return publish(b);
}

static void devcache_drop(const parasail_matrix_t *m);

// src/matrix/mod.rs:222-242 (Rust checks the index range and the builtin flag first)
extern "C" void parasail_matrix_set_value(parasail_matrix_t *matrix, int row, int col, int value)
{
    if (!matrix || !matrix->user_matrix) return;
    if (row < 0 || col < 0 || row >= matrix->length || col >= matrix->size) return;
    matrix->user_matrix[(size_t)row * matrix->size + col] = value;
    if (value > matrix->max) matrix->max = value;
    if (value < matrix->min) matrix->min = value;
    devcache_drop(matrix);
}

// src/matrix/mod.rs:297-307
extern "C" void parasail_matrix_free(parasail_matrix_t *matrix)
{
    if (!matrix) return;
    MatrixBox *b = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mx_mutex);
        auto it = g_boxes.find(matrix);
        if (it == g_boxes.end()) return;       // builtin or unknown: never freed
        b = it->second; g_boxes.erase(it);
    }
    devcache_drop(matrix);
    delete b;
}

// ===================================================================== device matrices ===
struct DevMat { PmxDevMatrix d; int16_t *scores; uint8_t *mapper; uint64_t hash; int rows; };
static std::mutex g_dev_mutex;
static std::unordered_map<uint64_t, DevMat> g_devmats;     // key = matrix ptr ^ device

static uint64_t mat_hash(const parasail_matrix_t *m)
{
    uint64_t h = 1469598103934665603ULL;
    auto mix = [&](int v) { h ^= (uint32_t)v; h *= 1099511628211ULL; };
    const size_t cells = (size_t)m->length * m->size;
    for (size_t i = 0; i < cells; ++i) mix(m->matrix[i]);
    for (int i = 0; i < 256; ++i) mix(m->mapper[i]);
    mix(m->size); mix(m->length); mix(m->type);
    return h;
}

static void devcache_drop(const parasail_matrix_t *m)
{
    std::lock_guard<std::mutex> lk(g_dev_mutex);
    for (auto it = g_devmats.begin(); it != g_devmats.end();) {
        if ((it->first >> 8) == ((uint64_t)(uintptr_t)m)) {
            (void)hipFree(it->second.scores); (void)hipFree(it->second.mapper);
            it = g_devmats.erase(it);
        } else ++it;
    }
}

// Upload (once per matrix content and device) the int16 score table + byte mapper.
static int get_devmat(const parasail_matrix_t *m, DevMat *out)
{
    if (!m || !m->matrix || !m->mapper || m->size <= 0 || m->size > 255) { set_err("bad matrix"); return -1; }
    if (m->max > 32767 || m->min < -32768) { set_err("matrix scores do not fit int16"); return -1; }
    int dev = 0; HIP_OR_RET(hipGetDevice(&dev));
    const uint64_t key = ((uint64_t)(uintptr_t)m << 8) | (uint64_t)(dev & 0xFF);
    const uint64_t h = mat_hash(m);
    std::lock_guard<std::mutex> lk(g_dev_mutex);
    auto it = g_devmats.find(key);
    if (it != g_devmats.end() && it->second.hash == h) { *out = it->second; return 0; }
    if (it != g_devmats.end()) { (void)hipFree(it->second.scores); (void)hipFree(it->second.mapper); g_devmats.erase(it); }
    const size_t cells = (size_t)m->length * m->size;
    std::vector<int16_t> s16(cells); std::vector<uint8_t> map8(256);
    for (size_t i = 0; i < cells; ++i) s16[i] = (int16_t)m->matrix[i];
    for (int i = 0; i < 256; ++i) {
        int v = m->mapper[i];
        if (v < 0 || v >= m->size) v = m->size - 1;
        map8[i] = (uint8_t)v;
    }
    DevMat dm; dm.hash = h; dm.rows = m->length;
    HIP_OR_RET(hipMalloc(&dm.scores, cells * 2));
    HIP_OR_RET(hipMalloc(&dm.mapper, 256));
    HIP_OR_RET(hipMemcpy(dm.scores, s16.data(), cells * 2, hipMemcpyHostToDevice));
    HIP_OR_RET(hipMemcpy(dm.mapper, map8.data(), 256, hipMemcpyHostToDevice));
    dm.d.scores = dm.scores; dm.d.mapper = dm.mapper; dm.d.msize = m->size; dm.d.min = m->min; dm.d.max = m->max;
    g_devmats[key] = dm;
    *out = dm;
    return 0;
}

// ============================================================================ results ====
enum {
    F_NW = 1 << 0, F_SG = 1 << 1, F_SW = 1 << 2, F_SATURATED = 1 << 6, F_BANDED = 1 << 7,
    F_SCAN = 1 << 10, F_STRIPED = 1 << 11, F_DIAG = 1 << 12, F_BLOCKED = 1 << 13,
    F_STATS = 1 << 16, F_TABLE = 1 << 17, F_ROWCOL = 1 << 18, F_TRACE = 1 << 19,
    F_BITS_8 = 1 << 20, F_BITS_16 = 1 << 21, F_BITS_32 = 1 << 22, F_BITS_64 = 1 << 23
};

struct PmxPendQueue;
struct parasail_result {
    int score, end_query, end_ref, flag;
    int matches, similar, length;
    int qlen, rlen;
    int *tables[4];        // score, matches, similar, length  [qlen*rlen]
    int *rows[4];          // [rlen]
    int *cols[4];          // [qlen]
    int8_t *trace;         // [qlen*rlen]
    // Deferred results (switch PMX_DEFER_ALIGN): non-null while the pair sits in a queue of calls that have not run yet; the first
    // accessor that needs a computed field (of ANY result of that queue) runs the whole queue as one batch.  A heap block holding a
    // shared reference to the queue, so that a result may be read or freed from another thread and outlives its creator thread.
    struct PmxPendRef *pending;
};
static void pend_resolve(const parasail_result_t *r);
#define RESOLVE(r) do { if (__atomic_load_n(&(r)->pending, __ATOMIC_ACQUIRE)) pend_resolve(r); } while (0)

extern "C" {
int parasail_result_get_score(const parasail_result_t *r) { RESOLVE(r); return r->score; }
int parasail_result_get_end_query(const parasail_result_t *r) { RESOLVE(r); return r->end_query; }
int parasail_result_get_end_ref(const parasail_result_t *r) { RESOLVE(r); return r->end_ref; }
int parasail_result_get_matches(const parasail_result_t *r) { RESOLVE(r); return r->matches; }
int parasail_result_get_similar(const parasail_result_t *r) { RESOLVE(r); return r->similar; }
int parasail_result_get_length(const parasail_result_t *r) { RESOLVE(r); return r->length; }
int *parasail_result_get_score_table(const parasail_result_t *r) { return r->tables[0]; }
int *parasail_result_get_matches_table(const parasail_result_t *r) { return r->tables[1]; }
int *parasail_result_get_similar_table(const parasail_result_t *r) { return r->tables[2]; }
int *parasail_result_get_length_table(const parasail_result_t *r) { return r->tables[3]; }
int *parasail_result_get_score_row(const parasail_result_t *r) { return r->rows[0]; }
int *parasail_result_get_matches_row(const parasail_result_t *r) { return r->rows[1]; }
int *parasail_result_get_similar_row(const parasail_result_t *r) { return r->rows[2]; }
int *parasail_result_get_length_row(const parasail_result_t *r) { return r->rows[3]; }
int *parasail_result_get_score_col(const parasail_result_t *r) { return r->cols[0]; }
int *parasail_result_get_matches_col(const parasail_result_t *r) { return r->cols[1]; }
int *parasail_result_get_similar_col(const parasail_result_t *r) { return r->cols[2]; }
int *parasail_result_get_length_col(const parasail_result_t *r) { return r->cols[3]; }
int *parasail_result_get_trace_table(const parasail_result_t *r) { return reinterpret_cast<int *>(r->trace); }
int parasail_result_is_nw(const parasail_result_t *r) { return !!(r->flag & F_NW); }
int parasail_result_is_sg(const parasail_result_t *r) { return !!(r->flag & F_SG); }
int parasail_result_is_sw(const parasail_result_t *r) { return !!(r->flag & F_SW); }
int parasail_result_is_saturated(const parasail_result_t *r) { RESOLVE(r); return !!(r->flag & F_SATURATED); }
int parasail_result_is_banded(const parasail_result_t *r) { return !!(r->flag & F_BANDED); }
int parasail_result_is_scan(const parasail_result_t *r) { return !!(r->flag & F_SCAN); }
int parasail_result_is_striped(const parasail_result_t *r) { return !!(r->flag & F_STRIPED); }
int parasail_result_is_diag(const parasail_result_t *r) { return !!(r->flag & F_DIAG); }
int parasail_result_is_blocked(const parasail_result_t *r) { return !!(r->flag & F_BLOCKED); }
int parasail_result_is_stats(const parasail_result_t *r) { return !!(r->flag & F_STATS); }
/* tests/test_parasail.rs:276-279, :397-399: stats+table -> is_table && is_stats && is_stats_table */
int parasail_result_is_stats_table(const parasail_result_t *r) { return (r->flag & F_STATS) && (r->flag & F_TABLE); }
int parasail_result_is_table(const parasail_result_t *r) { return !!(r->flag & F_TABLE); }
int parasail_result_is_rowcol(const parasail_result_t *r) { return !!(r->flag & F_ROWCOL); }
int parasail_result_is_stats_rowcol(const parasail_result_t *r) { return (r->flag & F_STATS) && (r->flag & F_ROWCOL); }
int parasail_result_is_trace(const parasail_result_t *r) { return !!(r->flag & F_TRACE); }

static void pend_cancel(parasail_result_t *r);
void parasail_result_free(parasail_result_t *r)
{
    if (!r) return;
    if (__atomic_load_n(&r->pending, __ATOMIC_ACQUIRE)) pend_cancel(r);      // a result that was never looked at: its pair leaves the queue
    for (int k = 0; k < 4; ++k) { free(r->tables[k]); free(r->rows[k]); free(r->cols[k]); }
    free(r->trace);
    free(r);
}
}  // extern "C"

// ====================================================================== single-pair run ===
struct RunSpec {
    int mode, sg_flags, band;
    int width;             // 0 sat, 8, 16, 32, 64
    bool stats, table, rowcol, trace;
    int vecflag;           // F_STRIPED / F_SCAN / F_DIAG / 0
};

static int run_batch_device(const pmx_config_t *cfg, int64_t n,
                            const uint8_t *d_qbuf, const int64_t *d_qoff, int q_shared,
                            const uint8_t *d_rbuf, const int64_t *d_roff,
                            int32_t max_qlen, int32_t max_rlen,
                            pmx_record_t *d_out, pmx_stats_t *d_stats_out, void *stream, int q_shared_wild = 1);

// One device block + one pinned host block per thread for the one-pair entry: a single H2D, the
// kernel(s), a single D2H.  (The reference's call is a CPU function of ~20 us; no allocation per call.)
struct SingleWs { void *dev = nullptr; void *pin = nullptr; size_t cap = 0; int device = -1; void *big = nullptr; size_t bigcap = 0; int bigdev = -1;
                  hipStream_t stream = nullptr; int streamdev = -1; };      // (a stream per host thread: one-pair calls of several threads overlap on the chip)
static thread_local SingleWs g_single;
// one device block per host thread for the one-pair calls that return tables (carved up per call: a dozen hipMalloc / hipFree
// per call cost more than the kernel)
// The block is kept between calls only up to SINGLE_BIG_KEEP: one 20 kbp x 20 kbp table call would otherwise pin gigabytes of HBM
// per host thread for the life of the process (the batch entries size their chunks from the free memory they find).
static const size_t SINGLE_BIG_KEEP = (size_t)256 << 20;
static void single_big_reserve(size_t bytes)
{
    int dev = 0; HIP_OR_DIE(hipGetDevice(&dev));
    if (g_single.bigdev == dev && g_single.bigcap >= bytes) return;
    if (g_single.big) (void)hipFree(g_single.big);
    g_single.big = nullptr; g_single.bigcap = 0; g_single.bigdev = -1;
    const size_t cap = bytes < (1u << 20) ? (1u << 20) : bytes + bytes / 2 <= SINGLE_BIG_KEEP ? bytes + bytes / 2 : bytes;
    HIP_OR_DIE(hipMalloc(&g_single.big, cap));
    g_single.bigcap = cap; g_single.bigdev = dev;
}
static void single_big_trim()
{
    if (g_single.big && g_single.bigcap > SINGLE_BIG_KEEP) {
        (void)hipFree(g_single.big);
        g_single.big = nullptr; g_single.bigcap = 0; g_single.bigdev = -1;
    }
}
static void single_reserve(size_t bytes)
{
    int dev = 0; HIP_OR_DIE(hipGetDevice(&dev));
    if (g_single.streamdev != dev) {
        // the reference's parallel story is threads calling align() side by side (tests/test_parasail.rs:689-723): on the legacy default
        // stream their launches would run one after the other (20 k pairs/s whatever the thread count, profiles/r04/thread_table.txt)
        if (g_single.stream) (void)hipStreamDestroy(g_single.stream);
        HIP_OR_DIE(hipStreamCreateWithFlags(&g_single.stream, hipStreamNonBlocking));
        g_single.streamdev = dev;
    }
    if (g_single.device == dev && g_single.cap >= bytes) return;
    if (g_single.dev) { (void)hipFree(g_single.dev); (void)hipHostFree(g_single.pin); }
    const size_t cap = bytes < 65536 ? 65536 : bytes * 2;
    HIP_OR_DIE(hipMalloc(&g_single.dev, cap));
    HIP_OR_DIE(hipHostMalloc(&g_single.pin, cap, hipHostMallocDefault));
    g_single.cap = cap; g_single.device = dev;
}

template <typename T> struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    void alloc(size_t n) { HIP_OR_DIE(hipMalloc(&p, (n ? n : 1) * sizeof(T))); }
    int try_alloc(size_t n) { return hipMalloc(&p, (n ? n : 1) * sizeof(T)) == hipSuccess ? 0 : -1; }   // entries that can report an error
};

// ---- deferred results (PMX_DEFER_ALIGN) ----------------------------------------------------------------------------------
// One queue per host thread and configuration: packed sequences as the batch entry wants them, and the results waiting for them.
struct PmxPendQueue {
    std::mutex mx;
    pmx_config_t cfg; RunSpec sp;
    std::vector<uint8_t> qbuf, rbuf;
    std::vector<int64_t> qoff{0}, roff{0};
    std::vector<parasail_result_t *> res;          // nullptr: the result was freed before anything asked for it
};
struct PmxPendRef { std::shared_ptr<PmxPendQueue> q; };
static thread_local std::shared_ptr<PmxPendQueue> g_pend;

extern "C" int pmx_align_batch(const pmx_config_t *cfg, int64_t n, const uint8_t *qbuf, const int64_t *qoff, const uint8_t *rbuf, const int64_t *roff,
                               pmx_record_t *out, pmx_stats_t *stats_out);

// runs the queue (its mutex held by the caller) and completes every result in it
static void pend_flush_locked(PmxPendQueue &q)
{
    const int64_t n = (int64_t)q.res.size();
    if (!n) return;
    std::vector<pmx_record_t> rec((size_t)n);
    std::vector<pmx_stats_t> st(q.sp.stats ? (size_t)n : 0);
    if (pmx_align_batch(&q.cfg, n, q.qbuf.data(), q.qoff.data(), q.rbuf.data(), q.roff.data(), rec.data(), q.sp.stats ? st.data() : nullptr))
        die(g_err, hipSuccess);
    for (int64_t k = 0; k < n; ++k) {
        parasail_result_t *r = q.res[(size_t)k];
        if (!r) continue;
        r->score = rec[(size_t)k].score; r->end_query = rec[(size_t)k].end_query; r->end_ref = rec[(size_t)k].end_ref;
        if (rec[(size_t)k].flags & PMX_FLAG_SATURATED) r->flag |= F_SATURATED;
        if (q.sp.stats) { r->matches = st[(size_t)k].matches; r->similar = st[(size_t)k].similar; r->length = st[(size_t)k].length; }
        PmxPendRef *ref = r->pending;
        __atomic_store_n(&r->pending, (PmxPendRef *)nullptr, __ATOMIC_RELEASE);
        delete ref;
    }
    q.qbuf.clear(); q.rbuf.clear(); q.qoff.assign(1, 0); q.roff.assign(1, 0); q.res.clear();
}
static void pend_resolve(const parasail_result_t *r)
{
    PmxPendRef *ref = __atomic_load_n(&r->pending, __ATOMIC_ACQUIRE);
    if (!ref) return;
    std::shared_ptr<PmxPendQueue> q = ref->q;          // (the flush deletes `ref`)
    std::lock_guard<std::mutex> lk(q->mx);
    if (__atomic_load_n(&r->pending, __ATOMIC_ACQUIRE)) pend_flush_locked(*q);
}
static void pend_cancel(parasail_result_t *r)
{
    PmxPendRef *ref = __atomic_load_n(&r->pending, __ATOMIC_ACQUIRE);
    if (!ref) return;
    std::shared_ptr<PmxPendQueue> q = ref->q;
    std::lock_guard<std::mutex> lk(q->mx);
    ref = __atomic_load_n(&r->pending, __ATOMIC_ACQUIRE);
    if (!ref) return;                                   // completed meanwhile
    for (auto &p : q->res) if (p == r) { p = nullptr; break; }     // its pair still runs with the batch; nobody reads the record
    __atomic_store_n(&r->pending, (PmxPendRef *)nullptr, __ATOMIC_RELEASE);
    delete ref;
}
// true: the pair was queued and `res` is pending
static bool pend_enqueue(parasail_result_t *res, const RunSpec &sp, const char *s1, int s1Len, const char *s2, int s2Len,
                         int open, int gap, const parasail_matrix_t *matrix)
{
    if (!pmx_env("PMX_DEFER_ALIGN")) return false;
    if ((long long)s1Len * s2Len > (1LL << 20)) return false;           // long pairs fill the chip on their own
    if (!g_pend) g_pend = std::make_shared<PmxPendQueue>();
    std::shared_ptr<PmxPendQueue> q = g_pend;
    std::lock_guard<std::mutex> lk(q->mx);
    const bool same = !q->res.empty() && q->sp.mode == sp.mode && q->sp.sg_flags == sp.sg_flags && q->sp.width == sp.width &&
                      q->sp.stats == sp.stats && q->cfg.open == open && q->cfg.extend == gap && q->cfg.matrix == matrix;
    if (!q->res.empty() && (!same || q->res.size() >= (1u << 18) || q->qbuf.size() + q->rbuf.size() > ((size_t)256 << 20)))
        pend_flush_locked(*q);                                           // another configuration, or enough for one batch
    if (q->res.empty()) {
        q->sp = sp; memset(&q->cfg, 0, sizeof q->cfg);
        q->cfg.mode = sp.mode; q->cfg.sg_flags = sp.sg_flags; q->cfg.open = open; q->cfg.extend = gap; q->cfg.width = sp.width;
        q->cfg.want = sp.stats ? PMX_WANT_STATS : 0; q->cfg.matrix = matrix;
    }
    q->qbuf.insert(q->qbuf.end(), (const uint8_t *)s1, (const uint8_t *)s1 + s1Len);
    q->rbuf.insert(q->rbuf.end(), (const uint8_t *)s2, (const uint8_t *)s2 + s2Len);
    q->qoff.push_back((int64_t)q->qbuf.size()); q->roff.push_back((int64_t)q->rbuf.size());
    q->res.push_back(res);
    PmxPendRef *ref = new PmxPendRef{q};
    __atomic_store_n(&res->pending, ref, __ATOMIC_RELEASE);
    return true;
}
// every pending result of the calling thread's queue, now (the mirrors call it when an aligner that deferred work goes away)
extern "C" void pmx_flush_deferred(void)
{
    if (!g_pend) return;
    std::shared_ptr<PmxPendQueue> q = g_pend;
    std::lock_guard<std::mutex> lk(q->mx);
    pend_flush_locked(*q);
}

static parasail_result_t *run_single(const RunSpec &sp, const char *s1, int s1Len, const char *s2, int s2Len,
                                     int open, int gap, const parasail_matrix_t *matrix)
{
    parasail_result_t *res = (parasail_result_t *)calloc(1, sizeof(parasail_result_t));
    if (!res) die("calloc", hipSuccess);
    res->qlen = s1Len; res->rlen = s2Len;
    int flag = sp.vecflag;
    flag |= sp.mode == PMX_MODE_NW ? F_NW : sp.mode == PMX_MODE_SG ? F_SG : F_SW;
    if (sp.band >= 0) flag |= F_BANDED;
    if (sp.stats) flag |= F_STATS;
    if (sp.table) flag |= F_TABLE;
    if (sp.rowcol) flag |= F_ROWCOL;
    if (sp.trace) flag |= F_TRACE;
    flag |= sp.width == 8 ? F_BITS_8 : sp.width == 16 ? F_BITS_16 : sp.width == 64 ? F_BITS_64 : F_BITS_32;
    res->flag = flag;
    if (!s1 || !s2 || s1Len <= 0 || s2Len <= 0 || !matrix) return res;   // degenerate: empty result, never NULL

    const bool pssm = matrix->type == PARASAIL_MATRIX_TYPE_PSSM;
    if (pssm && matrix->length != s1Len) die("PSSM length differs from the query length", hipSuccess);
    // Deferred results (PMX_DEFER_ALIGN): the reference's parallel story is user threads calling align() one pair at a time
    // (tests/test_parasail.rs:689-723); one launch + one wait per 150 x 150 pair is 50 us -- slower than one CPU core.  With the
    // switch on, the call only queues the pair and hands back a PENDING result; the first accessor runs the thread's queue as ONE batch.
    if (!sp.table && !sp.rowcol && !sp.trace && sp.band < 0 && !pssm && pend_enqueue(res, sp, s1, s1Len, s2, s2Len, open, gap, matrix)) return res;
    DevMat dm;
    if (get_devmat(matrix, &dm)) die(g_err, hipSuccess);

    if (!sp.table && !sp.rowcol && !sp.trace && sp.band < 0 && !pssm) {
        // score (+ stats) only: a one-pair batch through the same dispatcher as pmx_align_batch_device
        const size_t qpad = ((size_t)s1Len + 7) & ~(size_t)7, rpad = ((size_t)s2Len + 7) & ~(size_t)7;
        const size_t in_bytes = qpad + rpad + 4 * sizeof(int64_t);
        const size_t total = in_bytes + 64;
        single_reserve(total);
        unsigned char *h = (unsigned char *)g_single.pin, *d = (unsigned char *)g_single.dev;
        memcpy(h, s1, (size_t)s1Len); memcpy(h + qpad, s2, (size_t)s2Len);
        const int64_t offs[4] = {0, s1Len, 0, s2Len};
        memcpy(h + qpad + rpad, offs, sizeof offs);
        // Short pairs: the kernel reads the page-locked staging block itself and writes its record there (the block is mapped into
        // the device's address space), so the call is one launch and one wait -- no copy commands in front of and behind it.
        const bool zero_copy = in_bytes <= 4096;
        if (zero_copy) d = h;
        else HIP_OR_DIE(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, g_single.stream));
        pmx_config_t cfg; memset(&cfg, 0, sizeof cfg);
        cfg.mode = sp.mode; cfg.sg_flags = sp.sg_flags; cfg.open = open; cfg.extend = gap; cfg.width = sp.width;
        cfg.want = sp.stats ? PMX_WANT_STATS : 0; cfg.matrix = matrix;
        pmx_record_t *drec = (pmx_record_t *)(d + in_bytes);
        pmx_stats_t *dst = (pmx_stats_t *)(d + in_bytes + 16);
        const int64_t *doff = (const int64_t *)(d + qpad + rpad);
        if (run_batch_device(&cfg, 1, d, doff, 0, d + qpad, doff + 2, s1Len, s2Len, drec, sp.stats ? dst : nullptr, g_single.stream))
            die(g_err, hipSuccess);
        if (!zero_copy) HIP_OR_DIE(hipMemcpyAsync(h + in_bytes, d + in_bytes, 32, hipMemcpyDeviceToHost, g_single.stream));
        HIP_OR_DIE(hipStreamSynchronize(g_single.stream));
        pmx_record_t rec; pmx_stats_t st;
        memcpy(&rec, h + in_bytes, sizeof rec); memcpy(&st, h + in_bytes + 16, sizeof st);
        res->score = rec.score; res->end_query = rec.end_query; res->end_ref = rec.end_ref;
        if (rec.flags & PMX_FLAG_SATURATED) res->flag |= F_SATURATED;
        if (sp.stats) { res->matches = st.matches; res->similar = st.similar; res->length = st.length; }
        return res;
    }

    const size_t cells = (size_t)s1Len * s2Len;
    const int ntab = sp.stats ? 4 : 1;
    const bool rs_fits = pmx_general_lds_fits(matrix->length, matrix->size, s2Len);
    // one device block, carved up: [q | r | offsets][record, stats][boundary row][mapped reference][tables][rows, columns][trace]
    struct Ptr { void *p = nullptr; };
    struct { uint8_t *p; } dq, dr, drs = {nullptr}; struct { int64_t *p; } doff; struct { pmx_record_t *p; } drec; struct { pmx_stats_t *p; } dst;
    struct { int32_t *p; } dbound, dtab[4] = {{nullptr}, {nullptr}, {nullptr}, {nullptr}}, drow[4] = {{nullptr}, {nullptr}, {nullptr}, {nullptr}},
                           dcol[4] = {{nullptr}, {nullptr}, {nullptr}, {nullptr}};
    struct { int8_t *p; } dtrace = {nullptr};
    const size_t qpad = ((size_t)s1Len + 7) & ~(size_t)7, rpad = ((size_t)s2Len + 7) & ~(size_t)7, in_bytes = qpad + rpad + 4 * sizeof(int64_t);
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t o_in = carve(in_bytes), o_rec = carve(64), o_bound = carve((size_t)8 * s2Len * sizeof(int32_t)),
                 o_rs = rs_fits ? 0 : carve((size_t)s2Len + 32);
    size_t o_tab[4] = {0, 0, 0, 0}, o_row[4] = {0, 0, 0, 0}, o_col[4] = {0, 0, 0, 0}, o_trace = 0;
    if (sp.table) for (int k = 0; k < ntab; ++k) o_tab[k] = carve(cells * sizeof(int32_t));
    if (sp.rowcol) for (int k = 0; k < ntab; ++k) { o_row[k] = carve((size_t)s2Len * sizeof(int32_t)); o_col[k] = carve((size_t)s1Len * sizeof(int32_t)); }
    if (sp.trace) o_trace = carve(cells);
    single_big_reserve(off);
    single_reserve(in_bytes + 64);
    unsigned char *D = (unsigned char *)g_single.big, *h = (unsigned char *)g_single.pin;
    dq.p = D + o_in; dr.p = D + o_in + qpad; doff.p = (int64_t *)(D + o_in + qpad + rpad);
    drec.p = (pmx_record_t *)(D + o_rec); dst.p = (pmx_stats_t *)(D + o_rec + 16);
    dbound.p = (int32_t *)(D + o_bound);
    if (!rs_fits) drs.p = D + o_rs;
    if (sp.table) for (int k = 0; k < ntab; ++k) dtab[k].p = (int32_t *)(D + o_tab[k]);
    if (sp.rowcol) for (int k = 0; k < ntab; ++k) { drow[k].p = (int32_t *)(D + o_row[k]); dcol[k].p = (int32_t *)(D + o_col[k]); }
    if (sp.trace) dtrace.p = (int8_t *)(D + o_trace);
    memcpy(h, s1, (size_t)s1Len); memcpy(h + qpad, s2, (size_t)s2Len);
    const int64_t offs[4] = {0, s1Len, 0, s2Len};
    memcpy(h + qpad + rpad, offs, sizeof offs);
    HIP_OR_DIE(hipMemcpyAsync(D + o_in, h, in_bytes, hipMemcpyHostToDevice, nullptr));

    PmxGeneralArgs a; memset(&a, 0, sizeof a);
    a.qbuf = dq.p; a.qoff = doff.p; a.rbuf = dr.p; a.roff = doff.p + 2; a.n = 1; a.max_rlen = s2Len;
    a.scores = dm.d.scores; a.mapper = dm.d.mapper; a.msize = dm.d.msize;
    a.mat_rows = matrix->length; a.pssm = pssm ? 1 : 0;
    a.mode = sp.mode; a.sg_flags = sp.sg_flags; a.open = open; a.ext = gap; a.band = sp.band;
    a.bits = sp.width; a.max_qlen = s1Len;
    a.bound = dbound.p; a.bound_stride = (long long)8 * s2Len;
    if (!rs_fits) { a.rs_scratch = drs.p; a.rs_stride = (long long)s2Len + 32; }
    a.rec = drec.p; a.stats = dst.p;
    a.score_table = dtab[0].p; a.matches_table = dtab[1].p; a.similar_table = dtab[2].p; a.length_table = dtab[3].p;
    a.score_row = drow[0].p; a.matches_row = drow[1].p; a.similar_row = drow[2].p; a.length_row = drow[3].p;
    a.score_col = dcol[0].p; a.matches_col = dcol[1].p; a.similar_col = dcol[2].p; a.length_col = dcol[3].p;
    a.trace_table = dtrace.p;

    int rc = 1;
    if ((sp.table || sp.rowcol) && !sp.stats && !sp.trace && sp.band < 0 && !pssm && sp.width != 8 && sp.width != 16)
        // score table / last row and column: the row-by-row kernel that writes at HBM speed (pmx_table.hip); widths 8 / 16 keep the
        // general kernel, which reports their saturation
        rc = pmx_launch_table(sp.mode, sp.sg_flags, open, gap, dm.d, 1, dq.p, doff.p, 0, dr.p, doff.p + 2, s1Len, s2Len,
                              nullptr, dtab[0].p, drow[0].p, dcol[0].p, drec.p, nullptr);
    if (rc < 0) die("table kernel launch failed", (hipError_t)(-rc));
    if (sp.trace && !sp.table && !sp.rowcol && !sp.stats && sp.band < 0 && !pssm && sp.width != 8 && sp.width != 16) {
        // trace table alone (use_trace + get_cigar / get_traceback_strings): the same row-by-row kernel writes the bytes
        rc = pmx_launch_table(sp.mode, sp.sg_flags, open, gap, dm.d, 1, dq.p, doff.p, 0, dr.p, doff.p + 2, s1Len, s2Len,
                              nullptr, nullptr, nullptr, nullptr, drec.p, nullptr, dtrace.p);
        if (rc < 0) die("table kernel launch failed", (hipError_t)(-rc));
    }
    if (rc == 1) rc = pmx_launch_general(a, sp.stats, nullptr);
    if (rc) die("general kernel launch failed (matrix too large for LDS?)", hipSuccess);
    pmx_record_t rec; pmx_stats_t st = {0, 0, 0};
    HIP_OR_DIE(hipMemcpyAsync(h + in_bytes, D + o_rec, 32, hipMemcpyDeviceToHost, nullptr));
    HIP_OR_DIE(hipStreamSynchronize(nullptr));
    memcpy(&rec, h + in_bytes, sizeof rec);
    if (sp.stats) memcpy(&st, h + in_bytes + 16, sizeof st);
    res->score = rec.score; res->end_query = rec.end_query; res->end_ref = rec.end_ref;
    if (rec.flags & PMX_FLAG_SATURATED) res->flag |= F_SATURATED;
    res->matches = st.matches; res->similar = st.similar; res->length = st.length;
    auto fetch = [&](int32_t *d, size_t n) -> int * {
        if (!d) return nullptr;
        int *h = (int *)malloc(n * sizeof(int));
        if (!h) die("malloc", hipSuccess);
        HIP_OR_DIE(hipMemcpy(h, d, n * sizeof(int), hipMemcpyDeviceToHost));
        return h;
    };
    for (int k = 0; k < 4; ++k) {
        res->tables[k] = fetch(dtab[k].p, cells);
        res->rows[k] = fetch(drow[k].p, s2Len);
        res->cols[k] = fetch(dcol[k].p, s1Len);
    }
    if (sp.trace) {
        res->trace = (int8_t *)malloc(cells);
        if (!res->trace) die("malloc", hipSuccess);
        HIP_OR_DIE(hipMemcpy(res->trace, dtrace.p, cells, hipMemcpyDeviceToHost));
    }
    single_big_trim();
    return res;
}

// ============================================================== dispatch-name grammar ====
// {mode}{sg_gaps}{trace}{stats}{table}{vec}{profile}_{width}     src/aligner/mod.rs:319-329
// id = ((mode*7 + out)*3 + vec)*5 + width
//   mode : 0 nw, 1 sw, 2 + q*4 + d  (q,d in {none,b,e,x}) for sg
//   out  : 0 -, 1 table, 2 rowcol, 3 stats, 4 stats_table, 5 stats_rowcol, 6 trace
//   vec  : 0 striped, 1 scan, 2 diag          width: 0 sat, 1 8, 2 16, 3 32, 4 64
static const int N_MODE = 18, N_OUT = 7, N_VEC = 3, N_WIDTH = 5;
static const int N_IDS = N_MODE * N_OUT * N_VEC * N_WIDTH;

static RunSpec spec_from_id(int id)
{
    RunSpec sp; memset(&sp, 0, sizeof sp);
    const int w = id % N_WIDTH; id /= N_WIDTH;
    const int v = id % N_VEC; id /= N_VEC;
    const int o = id % N_OUT; id /= N_OUT;
    const int m = id;
    static const int widths[5] = {0, 8, 16, 32, 64};
    sp.width = widths[w];
    sp.vecflag = v == 0 ? F_STRIPED : v == 1 ? F_SCAN : F_DIAG;
    sp.table = (o == 1 || o == 4); sp.rowcol = (o == 2 || o == 5);
    sp.stats = (o >= 3 && o <= 5); sp.trace = (o == 6);
    sp.band = -1;
    if (m == 0) sp.mode = PMX_MODE_NW;
    else if (m == 1) sp.mode = PMX_MODE_SW;
    else {
        sp.mode = PMX_MODE_SG;
        const int qi = (m - 2) / 4, di = (m - 2) % 4;
        int f = 0;
        if (qi == 1 || qi == 3) f |= PMX_SG_QB;
        if (qi == 2 || qi == 3) f |= PMX_SG_QE;
        if (di == 1 || di == 3) f |= PMX_SG_DB;
        if (di == 2 || di == 3) f |= PMX_SG_DE;
        if (qi == 0 && di == 0) f = PMX_SG_ALL;      // plain "sg": every end free
        sp.sg_flags = f;
    }
    return sp;
}

static bool eat(const char *&p, const char *tok)
{
    const size_t n = strlen(tok);
    if (strncmp(p, tok, n) == 0) { p += n; return true; }
    return false;
}

// returns id or -1; *is_profile reports the _profile slot
static int parse_name(const char *name, bool *is_profile)
{
    if (!name) return -1;
    const char *p = name;
    eat(p, "parasail_");
    int m;
    if (eat(p, "nw")) m = 0;
    else if (eat(p, "sw")) m = 1;
    else if (eat(p, "sg")) {
        int qi = 0, di = 0;
        if (eat(p, "_qb")) qi = 1; else if (eat(p, "_qe")) qi = 2; else if (eat(p, "_qx")) qi = 3;
        if (eat(p, "_db")) di = 1; else if (eat(p, "_de")) di = 2; else if (eat(p, "_dx")) di = 3;
        m = 2 + qi * 4 + di;
    } else return -1;
    const bool trace = eat(p, "_trace");
    const bool stats = eat(p, "_stats");
    const bool table = eat(p, "_table");
    const bool rowcol = !table && eat(p, "_rowcol");
    if (trace && (stats || table || rowcol)) return -1;
    int o = trace ? 6 : stats ? (table ? 4 : rowcol ? 5 : 3) : (table ? 1 : rowcol ? 2 : 0);
    int v;
    if (eat(p, "_striped")) v = 0; else if (eat(p, "_scan")) v = 1; else if (eat(p, "_diag")) v = 2; else return -1;
    *is_profile = eat(p, "_profile");
    if (*is_profile && v == 2) return -1;
    int w;
    if (!strcmp(p, "_sat")) w = 0; else if (!strcmp(p, "_8")) w = 1; else if (!strcmp(p, "_16")) w = 2;
    else if (!strcmp(p, "_32")) w = 3; else if (!strcmp(p, "_64")) w = 4; else return -1;
    return ((m * N_OUT + o) * N_VEC + v) * N_WIDTH + w;
}

// ---- profiles --------------------------------------------------------------------------
struct parasail_profile {
    char *s1; int s1Len;
    const parasail_matrix_t *matrix;
    int stats; int width;
    // device copies of the query for the batch entries: one per device, uploaded on first use there, freed with the profile
    // (a Profile is Send + Sync in the reference, src/profile/mod.rs:392-395: threads driving different GPUs may share one)
    mutable std::vector<std::pair<int, uint8_t *>> *d_copies; mutable std::mutex *mx;
};

static parasail_profile_t *profile_new(const char *s1, int s1Len, const parasail_matrix_t *matrix, int stats, int width)
{
    if (!s1 || s1Len <= 0 || !matrix) return nullptr;       // -> Error::NullProfile (src/profile/mod.rs:101-103)
    parasail_profile_t *p = (parasail_profile_t *)calloc(1, sizeof *p);
    if (!p) return nullptr;
    p->s1 = (char *)malloc((size_t)s1Len + 1);
    if (!p->s1) { free(p); return nullptr; }
    memcpy(p->s1, s1, (size_t)s1Len); p->s1[s1Len] = 0;
    p->s1Len = s1Len; p->matrix = matrix; p->stats = stats; p->width = width;
    p->d_copies = new std::vector<std::pair<int, uint8_t *>>; p->mx = new std::mutex;
    return p;
}
extern "C" void parasail_profile_free(parasail_profile_t *p)
{
    if (!p) return;
    int cur = 0; const bool have_dev = hipGetDevice(&cur) == hipSuccess;
    for (auto &c : *p->d_copies) { if (have_dev) (void)hipSetDevice(c.first); (void)hipFree(c.second); }
    if (have_dev && !p->d_copies->empty()) (void)hipSetDevice(cur);
    delete p->d_copies; delete p->mx;
    free(p->s1); free(p);
}
// the query maps to a column beyond the first four of the matrix alphabet somewhere (the perm-table kernel cannot express that)
static int profile_has_wildcard(const parasail_profile_t *p)
{
    for (int i = 0; i < p->s1Len; ++i)
        if (p->matrix->mapper[(unsigned char)p->s1[i]] >= 4) return 1;
    return 0;
}

static int profile_device_query(const parasail_profile_t *p, const uint8_t **out)
{
    int dev = 0; HIP_OR_RET(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(*p->mx);
    for (auto &c : *p->d_copies) if (c.first == dev) { *out = c.second; return 0; }
    uint8_t *d = nullptr;
    HIP_OR_RET(hipMalloc(&d, (size_t)p->s1Len + 16));
    hipError_t e = hipMemcpy(d, p->s1, (size_t)p->s1Len, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); set_err("hipMemcpy: %s", hipGetErrorString(e)); return -(int)e; }
    p->d_copies->emplace_back(dev, d);
    *out = d;
    return 0;
}

#define PMX_DEFINE_PROFILE_CREATORS(ISA)                                                                    \
    extern "C" parasail_profile_t *parasail_profile_create##ISA##_sat(const char *s, const int n, const parasail_matrix_t *m) { return profile_new(s, n, m, 0, 0); }  \
    extern "C" parasail_profile_t *parasail_profile_create##ISA##_8(const char *s, const int n, const parasail_matrix_t *m) { return profile_new(s, n, m, 0, 8); }    \
    extern "C" parasail_profile_t *parasail_profile_create##ISA##_16(const char *s, const int n, const parasail_matrix_t *m) { return profile_new(s, n, m, 0, 16); }  \
    extern "C" parasail_profile_t *parasail_profile_create##ISA##_32(const char *s, const int n, const parasail_matrix_t *m) { return profile_new(s, n, m, 0, 32); }  \
    extern "C" parasail_profile_t *parasail_profile_create##ISA##_64(const char *s, const int n, const parasail_matrix_t *m) { return profile_new(s, n, m, 0, 64); }  \
    extern "C" parasail_profile_t *parasail_profile_create_stats##ISA##_sat(const char *s, const int n, const parasail_matrix_t *m) { return profile_new(s, n, m, 1, 0); }  \
    extern "C" parasail_profile_t *parasail_profile_create_stats##ISA##_8(const char *s, const int n, const parasail_matrix_t *m) { return profile_new(s, n, m, 1, 8); }    \
    extern "C" parasail_profile_t *parasail_profile_create_stats##ISA##_16(const char *s, const int n, const parasail_matrix_t *m) { return profile_new(s, n, m, 1, 16); }  \
    extern "C" parasail_profile_t *parasail_profile_create_stats##ISA##_32(const char *s, const int n, const parasail_matrix_t *m) { return profile_new(s, n, m, 1, 32); }  \
    extern "C" parasail_profile_t *parasail_profile_create_stats##ISA##_64(const char *s, const int n, const parasail_matrix_t *m) { return profile_new(s, n, m, 1, 64); }
PMX_DEFINE_PROFILE_CREATORS()
PMX_DEFINE_PROFILE_CREATORS(_sse_128)
PMX_DEFINE_PROFILE_CREATORS(_avx_256)
PMX_DEFINE_PROFILE_CREATORS(_neon_128)
PMX_DEFINE_PROFILE_CREATORS(_altivec_128)

// ---- one trampoline per dispatch name (a C function pointer carries no closure) ----------
template <int ID>
static parasail_result_t *tramp_f(const char *s1, const int s1Len, const char *s2, const int s2Len,
                                  const int open, const int gap, const parasail_matrix_t *matrix)
{
    return run_single(spec_from_id(ID), s1, s1Len, s2, s2Len, open, gap, matrix);
}
template <int ID>
static parasail_result_t *tramp_p(const parasail_profile_t *profile, const char *s2, const int s2Len,
                                  const int open, const int gap)
{
    if (!profile) die("NULL profile passed to a profile alignment function", hipSuccess);
    return run_single(spec_from_id(ID), profile->s1, profile->s1Len, s2, s2Len, open, gap, profile->matrix);
}
template <size_t... Is>
static parasail_function_t *const *make_ftable(std::index_sequence<Is...>)
{
    static parasail_function_t *const t[] = {&tramp_f<(int)Is>...};
    return t;
}
template <size_t... Is>
static parasail_pfunction_t *const *make_ptable(std::index_sequence<Is...>)
{
    static parasail_pfunction_t *const t[] = {&tramp_p<(int)Is>...};
    return t;
}

// src/aligner/mod.rs:345: NULL -> build() panics "Parasail function: {}, not found." (:353-358)
extern "C" parasail_function_t *parasail_lookup_function(const char *funcname)
{
    bool prof = false;
    const int id = parse_name(funcname, &prof);
    if (id < 0 || prof) return nullptr;
    return make_ftable(std::make_index_sequence<N_IDS>{})[id];
}
// src/aligner/mod.rs:349
extern "C" parasail_pfunction_t *parasail_lookup_pfunction(const char *funcname)
{
    bool prof = false;
    const int id = parse_name(funcname, &prof);
    if (id < 0 || !prof) return nullptr;
    return make_ptable(std::make_index_sequence<N_IDS>{})[id];
}

extern "C" int pmx_align_batch_banded(const pmx_config_t *cfg, const parasail_profile_t *profile, int64_t n,
                                      const uint8_t *qbuf, const int64_t *qoff, const uint8_t *rbuf, const int64_t *roff,
                                      int32_t band, const int32_t *diag, pmx_record_t *out);
// src/aligner/mod.rs:470-481.  Cells with |i-j| > k are excluded.
extern "C" parasail_result_t *parasail_nw_banded(const char *s1, const int s1Len, const char *s2, const int s2Len,
                                                 const int open, const int gap, const int k,
                                                 const parasail_matrix_t *matrix)
{
    RunSpec sp; memset(&sp, 0, sizeof sp);
    sp.mode = PMX_MODE_NW; sp.band = k < 0 ? 0 : k; sp.width = 32; sp.vecflag = 0;
    if (s1 && s2 && s1Len > 0 && s2Len > 0 && matrix && matrix->type == PARASAIL_MATRIX_TYPE_SQUARE && sp.band <= 63 &&
        matrix->size <= PMX_MAX_FAST_MSIZE) {
        // the band-only kernel: O(qlen * k) cells, no length limit ("for aligning large sequences", src/aligner/mod.rs:454-456)
        parasail_result_t *res = (parasail_result_t *)calloc(1, sizeof(parasail_result_t));
        if (!res) die("calloc", hipSuccess);
        res->qlen = s1Len; res->rlen = s2Len;
        res->flag = F_NW | F_BANDED | F_BITS_32;
        pmx_config_t cfg; memset(&cfg, 0, sizeof cfg);
        cfg.mode = PMX_MODE_NW; cfg.open = open; cfg.extend = gap; cfg.width = 32; cfg.matrix = matrix;
        const int64_t qo[2] = {0, s1Len}, ro[2] = {0, s2Len};
        pmx_record_t rec;
        if (pmx_align_batch_banded(&cfg, nullptr, 1, (const uint8_t *)s1, qo, (const uint8_t *)s2, ro, sp.band, nullptr, &rec))
            die(g_err, hipSuccess);
        res->score = rec.score; res->end_query = rec.end_query; res->end_ref = rec.end_ref;
        return res;
    }
    return run_single(sp, s1, s1Len, s2, s2Len, open, gap, matrix);
}

// ================================================================ traceback / CIGAR =====
// Host-side O(qlen+rlen) walk over the trace table the GPU produced (the reference's
// counterpart also runs on the host inside libparasail: src/alignment/mod.rs:390-419).
// State INS (E: consumes a reference char) prints 'D', state DEL (F: consumes a query char)
// prints 'I'  -- SAM sense with query = s1, reference = s2.  [UNPINNED, see DESIGN.md]
static const char CIG_INS_STATE = PMX_CIGAR_LETTER_FOR_INS_STATE, CIG_DEL_STATE = PMX_CIGAR_LETTER_FOR_DEL_STATE;   // include/pmx_conventions.h

static std::string walk_ops(const parasail_result_t *res, const char *seqA, int lena, const char *seqB, int lenb,
                            const parasail_matrix_t *matrix, int *beg_query, int *beg_ref)
{
    std::string rev;
    int i = res->end_query, j = res->end_ref;
    const bool sw = res->flag & F_SW, sg = res->flag & F_SG;
    if (sg) {
        if (i + 1 == lena) for (int k = lenb - 1; k > j; --k) rev.push_back(CIG_INS_STATE);
        else if (j + 1 == lenb) for (int k = lena - 1; k > i; --k) rev.push_back(CIG_DEL_STATE);
    }
    int where = PARASAIL_DIAG;
    while (i >= 0 || j >= 0) {
        if (i < 0) { if (sw) break; rev.push_back(CIG_INS_STATE); --j; continue; }
        if (j < 0) { if (sw) break; rev.push_back(CIG_DEL_STATE); --i; continue; }
        const int t = res->trace[(size_t)i * lenb + j];
        if (where == PARASAIL_DIAG) {
            if (t & PARASAIL_DIAG) {
                const bool eq = matrix->mapper[(unsigned char)seqA[i]] == matrix->mapper[(unsigned char)seqB[j]];
                rev.push_back(eq ? '=' : 'X'); --i; --j;
            } else if (t & PARASAIL_INS) where = PARASAIL_INS;
            else if (t & PARASAIL_DEL) where = PARASAIL_DEL;
            else break;
        } else if (where == PARASAIL_INS) {
            rev.push_back(CIG_INS_STATE);
            if (t & PARASAIL_DIAG_E) where = PARASAIL_DIAG;
            --j;
        } else {
            rev.push_back(CIG_DEL_STATE);
            if (t & PARASAIL_DIAG_F) where = PARASAIL_DIAG;
            --i;
        }
    }
    *beg_query = i + 1; *beg_ref = j + 1;
    return std::string(rev.rbegin(), rev.rend());
}

static const char BAM_OPS[] = "MIDNSHP=X";
// The letters of the two gap states are not pinned by anything the reference holds (include/pmx_conventions.h).  The compiled
// default can be exchanged at run time, without a rebuild, by a caller who holds real parasail output that says otherwise:
// PMX_CIGAR_SWAP_ID=1 swaps I and D in everything handed out (packed ops of get_cigar / ssw, decoded text, batch CIGAR text).
int pmx_cigar_swapped()
{
    const char *v = pmx_env("PMX_CIGAR_SWAP_ID");
    return v && *v && strcmp(v, "0") != 0;
}

extern "C" parasail_cigar_t *parasail_result_get_cigar(parasail_result_t *result, const char *seqA, int lena,
                                                       const char *seqB, int lenb, const parasail_matrix_t *matrix)
{
    if (!result || !result->trace || !matrix || lena != result->qlen || lenb != result->rlen) return nullptr;
    parasail_cigar_t *c = (parasail_cigar_t *)calloc(1, sizeof *c);
    if (!c) return nullptr;
    const std::string ops = walk_ops(result, seqA, lena, seqB, lenb, matrix, &c->beg_query, &c->beg_ref);
    c->seq = (uint32_t *)malloc(sizeof(uint32_t) * (ops.size() + 1));
    if (!c->seq) { free(c); return nullptr; }
    size_t k = 0; int n = 0;
    const int swap = pmx_cigar_swapped();
    while (k < ops.size()) {
        size_t run = 1;
        while (k + run < ops.size() && ops[k + run] == ops[k]) ++run;
        uint32_t op = (uint32_t)(strchr(BAM_OPS, ops[k]) - BAM_OPS);
        if (swap && (op == 1u || op == 2u)) op ^= 3u;             // I (1) <-> D (2)
        c->seq[n++] = ((uint32_t)run << 4) | op;
        k += run;
    }
    c->len = n;
    return c;
}

// src/alignment/mod.rs:410: the returned string is adopted by Rust with CString::from_raw
// and must be a plain malloc block.
extern "C" char *parasail_cigar_decode(parasail_cigar_t *cigar)
{
    if (!cigar) return nullptr;
    std::string s;
    for (int k = 0; k < cigar->len; ++k) {
        s += std::to_string(cigar->seq[k] >> 4);
        s.push_back(BAM_OPS[cigar->seq[k] & 0xF]);
    }
    char *out = (char *)malloc(s.size() + 1);
    if (out) memcpy(out, s.c_str(), s.size() + 1);
    return out;
}
extern "C" void parasail_cigar_free(parasail_cigar_t *cigar) { if (cigar) { free(cigar->seq); free(cigar); } }

// src/alignment/mod.rs:356-376: the three strings are malloc blocks adopted by Rust.
extern "C" parasail_traceback_t *parasail_result_get_traceback(parasail_result_t *result, const char *seqA, int lena,
        const char *seqB, int lenb, const parasail_matrix_t *matrix, char match, char pos, char neg)
{
    if (!result || !result->trace || !matrix || lena != result->qlen || lenb != result->rlen) return nullptr;
    int bq = 0, br = 0;
    const std::string ops = walk_ops(result, seqA, lena, seqB, lenb, matrix, &bq, &br);
    const size_t n = ops.size();
    parasail_traceback_t *tb = (parasail_traceback_t *)calloc(1, sizeof *tb);
    if (!tb) return nullptr;
    tb->query = (char *)malloc(n + 1); tb->comp = (char *)malloc(n + 1); tb->ref = (char *)malloc(n + 1);
    if (!tb->query || !tb->comp || !tb->ref) { free(tb->query); free(tb->comp); free(tb->ref); free(tb); return nullptr; }
    int i = bq, j = br;
    for (size_t k = 0; k < n; ++k) {
        const char o = ops[k];
        if (o == '=' || o == 'X') {
            const int s = matrix->matrix[(size_t)matrix->size *
                              (matrix->type == PARASAIL_MATRIX_TYPE_PSSM ? i : matrix->mapper[(unsigned char)seqA[i]]) +
                              matrix->mapper[(unsigned char)seqB[j]]];
            tb->query[k] = seqA[i]; tb->ref[k] = seqB[j];
            tb->comp[k] = (o == '=') ? match : (s > 0 ? pos : neg);
            ++i; ++j;
        } else if (o == CIG_INS_STATE) { tb->query[k] = '-'; tb->ref[k] = seqB[j]; tb->comp[k] = ' '; ++j; }
        else { tb->query[k] = seqA[i]; tb->ref[k] = '-'; tb->comp[k] = ' '; ++i; }
    }
    tb->query[n] = tb->comp[n] = tb->ref[n] = 0;
    return tb;
}
extern "C" void parasail_traceback_free(parasail_traceback_t *tb)
{
    if (tb) { free(tb->query); free(tb->comp); free(tb->ref); free(tb); }
}

// src/alignment/mod.rs:310-344 (print_traceback): blocks of `width` columns, names padded to
// name_width, optional summary line.
extern "C" void parasail_traceback_generic(const char *seqA, int lena, const char *seqB, int lenb,
        const char *nameA, const char *nameB, const parasail_matrix_t *matrix, parasail_result_t *result,
        char match, char pos, char neg, int width, int name_width, int use_stats)
{
    parasail_traceback_t *tb = parasail_result_get_traceback(result, seqA, lena, seqB, lenb, matrix, match, pos, neg);
    if (!tb) { printf("(no traceback available)\n"); return; }
    const int n = (int)strlen(tb->query);
    if (width <= 0) width = 80;
    int bq = 0, br = 0;
    (void)walk_ops(result, seqA, lena, seqB, lenb, matrix, &bq, &br);
    int qi = bq, ri = br, nmatch = 0, ngap = 0, nmis = 0;
    for (int k = 0; k < n; k += width) {
        const int w = (n - k < width) ? n - k : width;
        int qadv = 0, radv = 0;
        for (int c = 0; c < w; ++c) {
            if (tb->query[k + c] != '-') ++qadv;
            if (tb->ref[k + c] != '-') ++radv;
            if (tb->query[k + c] == '-' || tb->ref[k + c] == '-') ++ngap;
            else if (tb->comp[k + c] == match) ++nmatch; else ++nmis;
        }
        printf("\n%*s %9d %.*s %9d\n", name_width, nameB ? nameB : "", ri + 1, w, tb->ref + k, ri + radv);
        printf("%*s %9s %.*s\n", name_width, "", "", w, tb->comp + k);
        printf("%*s %9d %.*s %9d\n", name_width, nameA ? nameA : "", qi + 1, w, tb->query + k, qi + qadv);
        qi += qadv; ri += radv;
    }
    if (use_stats) {
        printf("\nLength: %d\nIdentity:   %d/%d\nMismatches: %d/%d\nGaps:       %d/%d\nScore: %d\n",
               n, nmatch, n, nmis, n, ngap, n, result->score);
    }
    parasail_traceback_free(tb);
}

// ===================================================================== SSW emulation ====
// src/aligner/mod.rs:491-529, src/alignment/mod.rs:506-551: local alignment with begin and
// end coordinates and a packed CIGAR.  Runs sw+trace on the GPU, walks the trace on the host.
extern "C" parasail_result_ssw_t *parasail_ssw(const char *s1, const int s1Len, const char *s2, const int s2Len,
                                               const int open, const int gap, const parasail_matrix_t *matrix)
{
    parasail_result_ssw_t *out = (parasail_result_ssw_t *)calloc(1, sizeof *out);
    if (!out) die("calloc", hipSuccess);
    RunSpec sp; memset(&sp, 0, sizeof sp);
    sp.mode = PMX_MODE_SW; sp.band = -1; sp.width = 32; sp.trace = true;
    parasail_result_t *r = run_single(sp, s1, s1Len, s2, s2Len, open, gap, matrix);
    if (r->trace) {
        parasail_cigar_t *c = parasail_result_get_cigar(r, s1, s1Len, s2, s2Len, matrix);
        out->score1 = (uint16_t)(r->score > 65535 ? 65535 : (r->score < 0 ? 0 : r->score));
        out->ref_end1 = r->end_ref; out->read_end1 = r->end_query;
        if (c) {
            out->ref_begin1 = c->beg_ref; out->read_begin1 = c->beg_query;
            out->cigar = c->seq; out->cigarLen = c->len;
            free(c);                   // the seq array is now owned by the ssw result
        }
    }
    parasail_result_free(r);
    return out;
}
extern "C" parasail_profile_t *parasail_ssw_init(const char *s1, const int s1Len, const parasail_matrix_t *matrix,
                                                 const int8_t score_size)
{
    (void)score_size;
    return profile_new(s1, s1Len, matrix, 1, 0);
}
extern "C" void parasail_result_ssw_free(parasail_result_ssw_t *r) { if (r) { free(r->cigar); free(r); } }

// ============================================================================ batches ===
static int check_cfg(const pmx_config_t *cfg)
{
    if (!cfg || !cfg->matrix) { set_err("null config or matrix"); return -1; }
    if (cfg->mode < 0 || cfg->mode > 2) { set_err("bad mode %d", cfg->mode); return -1; }
    if (cfg->width != 0 && cfg->width != 8 && cfg->width != 16 && cfg->width != 32 && cfg->width != 64) {
        set_err("bad width %d", cfg->width); return -1;
    }
    if (cfg->open < 0 || cfg->extend < 0) { set_err("gap penalties are passed as positive numbers"); return -1; }
    return 0;
}

static bool fast_sw_eligible(const pmx_config_t *cfg)
{
    // (width 8 included: for local alignment the saturation rule only needs the score, see PmxBatch::sat_above)
    return cfg->mode == PMX_MODE_SW && (cfg->want & ~PMX_WANT_SORTED) == 0 &&
           cfg->matrix->type == PARASAIL_MATRIX_TYPE_SQUARE;
}

extern "C" const char *pmx_kernel_for(const pmx_config_t *cfg, int32_t max_qlen, int32_t max_rlen)
{
    if (check_cfg(cfg)) return "invalid";
    if (fast_sw_eligible(cfg) && cfg->matrix->size <= PMX_MAX_FAST_MSIZE && max_qlen <= 2048 && max_rlen <= 60000)
        return "pmx_sw16_kernel";
    if ((cfg->mode == PMX_MODE_NW || cfg->mode == PMX_MODE_SG) && (cfg->want & ~PMX_WANT_SORTED) == 0 &&
        cfg->matrix->type == PARASAIL_MATRIX_TYPE_SQUARE && cfg->open >= cfg->extend && max_qlen <= 2048 &&
        cfg->matrix->size < PMX_MAX_FAST_MSIZE)
        return "pmx_nwsg16_kernel";
    if ((cfg->mode == PMX_MODE_NW || cfg->mode == PMX_MODE_SG) && (cfg->want & ~PMX_WANT_SORTED) == PMX_WANT_STATS && cfg->width != 8 &&
        cfg->matrix->type == PARASAIL_MATRIX_TYPE_SQUARE && cfg->open >= cfg->extend && cfg->extend >= 1 &&
        max_qlen <= 1024 && cfg->matrix->size < PMX_MAX_FAST_MSIZE)
        return "pmx_stats16_kernel";
    if ((cfg->want & ~PMX_WANT_SORTED) == 0 && cfg->matrix->type == PARASAIL_MATRIX_TYPE_SQUARE && cfg->matrix->size <= 64 &&
        (cfg->mode == PMX_MODE_SW || (cfg->width != 8 && cfg->width != 16)))
        return "pmx_long32_kernel";
    return "pmx_general_kernel";
}

// grow-only per-(thread,device) scratch for the general kernel's band boundary rows
struct Scratch { void *p = nullptr; size_t cap = 0; int dev = -1; };
enum { SCR_BOUND = 0, SCR_TRACE = 1, SCR_OPS = 2, SCR_SORT = 3, SCR_RETRY = 4,
       SCR_HQ = 5, SCR_HR = 6, SCR_HQO = 7, SCR_HRO = 8, SCR_HREC = 9, SCR_HST = 10,      // staging of the host-buffer batch entry
       SCR_CIG = 11,                                                                     // device CIGAR entry: counts, begins, text lengths, scan scratch
       SCR_HTEXT = 12, SCR_HTOFF = 13,                                                   // staging of the host CIGAR entry
       SCR_HQ2 = 14, SCR_HR2 = 15,                                                       // 2-bit packed input as it arrived
       SCR_LONG = 16,                                                                    // boundary granules + band candidates of pmx_long.hip
       SCR_SLOTS = 17 };
static thread_local Scratch g_scratch_pool[SCR_SLOTS];
static int scratch_reserve(size_t bytes, void **out, int slot = SCR_BOUND)
{
    Scratch &g_scratch = g_scratch_pool[slot];
    int dev = 0; HIP_OR_RET(hipGetDevice(&dev));
    if (g_scratch.dev != dev || g_scratch.cap < bytes) {
        if (g_scratch.p) {                                     // (a block of another device is released on that device)
            if (g_scratch.dev != dev) (void)hipSetDevice(g_scratch.dev);
            (void)hipFree(g_scratch.p);
            if (g_scratch.dev != dev) (void)hipSetDevice(dev);
        }
        g_scratch.p = nullptr; g_scratch.cap = 0; g_scratch.dev = dev;
        HIP_OR_RET(hipMalloc(&g_scratch.p, bytes ? bytes : 16));
        g_scratch.cap = bytes;
    }
    *out = g_scratch.p;
    return 0;
}

static thread_local const char *g_last_kernel = "";
extern "C" const char *pmx_last_kernel(void) { return g_last_kernel; }


// General kernel (one wave per pair) over a batch.  Its scratch -- the boundary row between 64-row bands, 8 ints per reference
// column, and for references beyond the LDS a mapped copy in HBM -- only has to cover the pairs of one launch: chunks of ~2 GB.
// band >= 0: cells with |(j - i) - diag[pair]| > band are excluded (diag == nullptr: the main diagonal).
static int general_batch(const pmx_config_t *cfg, const DevMat &dm, int64_t n,
                         const uint8_t *d_qbuf, const int64_t *d_qoff, int q_shared,
                         const uint8_t *d_rbuf, const int64_t *d_roff, int32_t max_rlen,
                         int band, const int32_t *d_diag, bool want_stats,
                         pmx_record_t *d_out, pmx_stats_t *d_stats_out, hipStream_t st, int32_t max_qlen = 0)
{
    const size_t stride = (size_t)8 * max_rlen;
    const bool fits = pmx_general_lds_fits(dm.d.msize, dm.d.msize, max_rlen);
    const size_t rs_stride = fits ? 0 : (((size_t)max_rlen + 8 + 15) & ~(size_t)15);
    const size_t per_pair = stride * sizeof(int32_t) + rs_stride;
    const char *cb = pmx_env("PMX_GENERAL_CHUNK_BYTES");
    int64_t chunk = (int64_t)((cb && atof(cb) > 0 ? atof(cb) : 2e9) / (double)per_pair);
    if (chunk < 1) chunk = 1;
    if (chunk > n) chunk = n;
    void *bound = nullptr;
    if (scratch_reserve((size_t)chunk * per_pair, &bound)) return -1;
    for (int64_t c0 = 0; c0 < n; c0 += chunk) {
        const int64_t m = (n - c0 < chunk) ? n - c0 : chunk;
        PmxGeneralArgs a; memset(&a, 0, sizeof a);
        a.qbuf = d_qbuf; a.qoff = q_shared ? nullptr : d_qoff + c0; a.shared_qlen = q_shared;
        a.rbuf = d_rbuf; a.roff = d_roff + c0; a.n = m; a.max_rlen = max_rlen;
        a.scores = dm.d.scores; a.mapper = dm.d.mapper; a.msize = dm.d.msize; a.mat_rows = dm.d.msize; a.pssm = 0;
        a.mode = cfg->mode; a.sg_flags = cfg->sg_flags; a.open = cfg->open; a.ext = cfg->extend;
        a.band = band; a.diag = d_diag ? d_diag + c0 : nullptr;
        a.bits = cfg->width; a.max_qlen = max_qlen;
        a.bound = (int32_t *)bound; a.bound_stride = (long long)stride;
        if (!fits) { a.rs_scratch = (uint8_t *)bound + (size_t)chunk * stride * sizeof(int32_t); a.rs_stride = (long long)rs_stride; }
        a.rec = d_out + c0; a.stats = d_stats_out ? d_stats_out + c0 : nullptr;
        const int rc = pmx_launch_general(a, want_stats, st);
        if (rc) { set_err("general kernel launch failed (%d)", rc); return rc < 0 ? rc : -1; }
    }
    return 0;
}

// ---- statistics of the profile arm by traceback ------------------------------------------------------------
// matches / similar / length are properties of the one path the coupled statistics tables follow (same decisions, same
// tie-breaks as the traceback bits): the shared-profile sweep writes the packed 4-bit records (14.75 instructions per two
// cells against 34 for the kernel that carries nine statistics planes) and the walk counts along the path.  Chunks bound
// the trace scratch; the walk of chunk c runs beside the sweep of chunk c + 1 on a second stream.
struct TraceWs { hipStream_t walk = nullptr, aux = nullptr, aux2 = nullptr; hipEvent_t sweep_done[3] = {nullptr, nullptr, nullptr}, walk_done[3] = {nullptr, nullptr, nullptr}, start = nullptr; int dev = -1; };
static thread_local TraceWs g_tws;
static int trace_ws_init()
{
    int dev = 0; HIP_OR_RET(hipGetDevice(&dev));
    if (g_tws.dev == dev) return 0;
    if (g_tws.walk) {                                   // the thread moved to another device: release the old device's objects
        (void)hipStreamDestroy(g_tws.walk); (void)hipStreamDestroy(g_tws.aux); (void)hipStreamDestroy(g_tws.aux2); (void)hipEventDestroy(g_tws.start);
        for (int k = 0; k < 3; ++k) { (void)hipEventDestroy(g_tws.sweep_done[k]); (void)hipEventDestroy(g_tws.walk_done[k]); }
        g_tws = TraceWs();
    }
    // the walk gets the higher priority: its few, latency-bound workgroups slip in between the sweep's as those retire
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    HIP_OR_RET(hipStreamCreateWithPriority(&g_tws.walk, hipStreamNonBlocking, prio_hi));
    HIP_OR_RET(hipStreamCreateWithFlags(&g_tws.aux, hipStreamNonBlocking));
    HIP_OR_RET(hipStreamCreateWithFlags(&g_tws.aux2, hipStreamNonBlocking));
    HIP_OR_RET(hipEventCreateWithFlags(&g_tws.start, hipEventDisableTiming));
    for (int k = 0; k < 3; ++k) {
        HIP_OR_RET(hipEventCreateWithFlags(&g_tws.sweep_done[k], hipEventDisableTiming));
        HIP_OR_RET(hipEventCreateWithFlags(&g_tws.walk_done[k], hipEventDisableTiming));
    }
    g_tws.dev = dev;
    return 0;
}

// 0 done (asynchronously on st), 1 not eligible, <0 error
// Upload progress of the calling host entry (pmx_align_profile_batch): reference slices still travelling on a copy stream.  A
// device routine that works through the batch in chunks of its own waits, per chunk, only for the slices that chunk reads.
struct UploadHook { int K = 0; int64_t hi[8]; hipEvent_t ev[8]; int waited[2] = {0, 0}; std::atomic<int> recorded{0}; std::atomic<int> failed{0}; };
static thread_local UploadHook *g_upload = nullptr;
static int upload_wait(int64_t upto /* references [0, upto) are about to be read */, hipStream_t st, int which /* 0 / 1: the stream's own progress */)
{
    UploadHook *u = g_upload;
    if (!u) return 0;
    int &w = u->waited[which];
    while (w < u->K && (w == 0 || u->hi[w - 1] < upto)) {
        while (u->recorded.load(std::memory_order_acquire) <= w && !u->failed.load()) std::this_thread::yield();   // (the uploading thread records the events)
        if (u->failed.load()) { set_err("upload of the references failed"); return -1; }
        HIP_OR_RET(hipStreamWaitEvent(st, u->ev[w], 0));
        ++w;
    }
    return 0;
}

static int stats_by_trace_shared(const pmx_config_t *cfg, const DevMat &dm, const PmxBatch &b,
                                 pmx_record_t *d_out, pmx_stats_t *d_stats, hipStream_t st)
{
    if (pmx_env("PMX_NO_STATS_BY_TRACE")) return 1;
    int variant = 0, Tmax = 0, G = 0, R = 0; size_t tbytes = 0;
    if (pmx_nwsgq_trace_plan(b, dm.d, cfg->mode, cfg->open, cfg->extend, &variant, &Tmax, &tbytes, &G, &R) != 0) return 1;
    if (trace_ws_init()) return -1;
    const long long NP = 2 * (64 / G) * 4;                     // pairs per workgroup of the sweep
    double chunk_bytes = 40e9;                                 // (measured on cfg 3: 8 GB chunks 55.0 ms, 24 GB 53.1 ms, 40 GB 51.0 ms)
    { size_t fb = 0, tb = 0; if (hipMemGetInfo(&fb, &tb) == hipSuccess && 0.2 * (double)fb < chunk_bytes) chunk_bytes = 0.2 * (double)fb; }
    if (const char *e = pmx_env("PMX_STATS_CHUNK_BYTES")) chunk_bytes = atof(e);      // tests force small chunks
    const double per_pair = (double)tbytes / (double)b.n;
    long long chunk = (long long)(chunk_bytes / per_pair) / NP * NP;
    if (chunk < NP) chunk = NP;
    bool by_rounds = false;
    if (chunk >= b.n) chunk = b.n;
    else {
        // Whole ROUNDS of resident workgroups per chunk: the waves of a sweep over equally long references all take the same time,
        // so a launch of N workgroups runs for ceil(N / resident) rounds -- cfg 3 in three equal chunks of 1 042 workgroups each
        // (512 resident) ran 5 + 3 rounds where 6.1 were needed: 45.2 -> 42.5 ms (the remainder chunk first instead of last: 42.8).
        // Otherwise (a round does not fit a chunk): equal shares.
        const long long round = pmx_env("PMX_STATS_EQUAL_CHUNKS") ? 0 : pmx_nwsgq_trace_round_pairs(variant, dm.d, cfg->mode, cfg->sg_flags);
        if (round > 0 && chunk >= round) {
            chunk = chunk / round * round; by_rounds = true;
            if (g_upload) chunk = round;                       // (host entry: the references arrive in slices -- a sweep starts as soon as ONE round's worth is up)
        }
        else {
            const long long nch = (b.n + chunk - 1) / chunk;
            chunk = ((b.n + nch - 1) / nch + NP - 1) / NP * NP;
        }
    }
    PmxBatch bc = b; bc.n = chunk;
    size_t cbytes = 0;
    (void)pmx_nwsgq_trace_plan(bc, dm.d, cfg->mode, cfg->open, cfg->extend, &variant, &Tmax, &cbytes, &G, &R);
    cbytes = (cbytes + 255) & ~(size_t)255;
    // (PMX_STATS_NO_OVERLAP: sweep and walk of every chunk back to back on the caller's stream, one trace buffer -- the form the
    //  serialised kernel traces under profiles/ are taken in: per-launch durations of overlapping launches cannot be added up)
    const bool two = chunk < b.n && !pmx_env("PMX_STATS_NO_OVERLAP");
    // The remainder after the whole rounds (a few long waves that would end the call with the chip nearly idle, after waiting for a trace
    // buffer to come free: 3.1 ms of a 38 ms cfg-3 step for 1.7 % of the pairs, measured) goes FIRST, on a stream and a trace buffer of
    // its own, beside the first big chunks.  Not through the host entry: there the last references are the last to arrive.
    long long rem_n = 0;
    if (by_rounds && two && !g_upload && b.n % chunk != 0 && !pmx_env("PMX_STATS_TAIL_LAST")) rem_n = b.n % chunk;
    size_t rbytes = 0;
    if (rem_n) {
        PmxBatch br = b; br.n = rem_n;
        int v_ = 0, T_ = 0, G_ = 0, R_ = 0;
        (void)pmx_nwsgq_trace_plan(br, dm.d, cfg->mode, cfg->open, cfg->extend, &v_, &T_, &rbytes, &G_, &R_);
        size_t r2 = 0;
        if (pmx_nwsgq_trace_plan(br, dm.d, cfg->mode, cfg->open, cfg->extend, &v_, &T_, &r2, &G_, &R_, 1) == 0 && r2 > rbytes) rbytes = r2;
        rbytes = (rbytes + 255) & ~(size_t)255;
    }
    uint32_t *tbuf = nullptr;
    if (scratch_reserve(cbytes * (two ? 2 : 1) + rbytes, (void **)&tbuf, SCR_TRACE)) return -1;
    const bool sg = cfg->mode == PMX_MODE_SG;
    const int col_pen = !(sg && (cfg->sg_flags & PMX_SG_QB)), row_pen = !(sg && (cfg->sg_flags & PMX_SG_DB));
    // Sweeps of consecutive chunks go to two streams in turn (the caller's and an internal one): a chunk is a few thousand equally
    // long waves, so the tail of chunk c's launch is backfilled by chunk c + 1's workgroups instead of idling the chip.
    if (two) { HIP_OR_RET(hipEventRecord(g_tws.start, st)); HIP_OR_RET(hipStreamWaitEvent(g_tws.aux, g_tws.start, 0)); }
    if (rem_n) HIP_OR_RET(hipStreamWaitEvent(g_tws.aux2, g_tws.start, 0));
    // one chunk: positions [c0, c0 + n_k) of the batch; sweep on `sws` into `tb`, walk on the walk stream behind it (slot = which events)
    auto run_chunk = [&](long long c0, long long n_k, uint32_t *tb, size_t tb_bytes, hipStream_t sws, int slot, bool short_waves) -> int {
        PmxBatch bk = b;
        bk.n = n_k;
        pmx_record_t *out_k = d_out; pmx_stats_t *st_k = d_stats;
        if (b.perm) bk.perm = b.perm + c0;                    // positions c0 .. of the processing order; records stay indexed by pair
        else { bk.roff = b.roff + c0; out_k = d_out + c0; st_k = d_stats + c0; }
        if (!b.perm && upload_wait(c0 + bk.n, sws, sws == st ? 0 : 1)) return -1;                  // (host entry: this chunk's references are up)
        // The remainder after the whole rounds is less than one round: it runs on the shape with half the rows per lane (<32,10> for
        // <16,20> / <16,19>) -- twice the waves, each half as long
        int variant_k = variant, Tmax_k = Tmax, G_k = G, R_k = R;
        if (short_waves && R >= 19 && !pmx_env("PMX_STATS_NO_SHORT_TAIL")) {
            int v2 = 0, T2 = 0, G2 = 0, R2 = 0; size_t tb2 = 0;
            if (pmx_nwsgq_trace_plan(bk, dm.d, cfg->mode, cfg->open, cfg->extend, &v2, &T2, &tb2, &G2, &R2, 1) == 0 && tb2 <= tb_bytes) {
                variant_k = v2; Tmax_k = T2; G_k = G2; R_k = R2;
            }
        }
        const int gsel_k = G_k == 16 ? 1 : G_k == 32 ? 2 : 3;
        int rc = pmx_launch_nwsgq_trace(variant_k, bk, dm.d, cfg->mode, cfg->sg_flags, cfg->open, cfg->extend, out_k, tb, Tmax_k, sws);
        if (rc) { set_err("shared-profile traceback sweep failed (%d)", rc); return rc < 0 ? rc : -1; }
        hipStream_t ws = st;
        if (two) {
            HIP_OR_RET(hipEventRecord(g_tws.sweep_done[slot], sws));
            HIP_OR_RET(hipStreamWaitEvent(g_tws.walk, g_tws.sweep_done[slot], 0));
            ws = g_tws.walk;
        }
        rc = pmx_launch_walkp(gsel_k, R_k, bk, dm.d, cfg->mode, cfg->open, cfg->extend, Tmax_k, 0, st_k, row_pen, col_pen,
                              tb, out_k, nullptr, nullptr, 0, nullptr, nullptr, nullptr, ws);
        if (rc) { set_err("statistics walk failed (%d)", rc); return rc < 0 ? rc : -1; }
        if (two) HIP_OR_RET(hipEventRecord(g_tws.walk_done[slot], ws));
        return 0;
    };
    if (rem_n) {
        const int rc = run_chunk(b.n - rem_n, rem_n, (uint32_t *)((unsigned char *)tbuf + 2 * cbytes), rbytes, g_tws.aux2, 2, true);
        if (rc) return rc;
    }
    int idx = 0;
    for (long long c0 = 0; c0 < b.n - rem_n; c0 += chunk, ++idx) {
        const long long n_k = (b.n - rem_n - c0 < chunk) ? b.n - rem_n - c0 : chunk;
        uint32_t *tb = (uint32_t *)((unsigned char *)tbuf + (two ? (size_t)(idx & 1) * cbytes : 0));
        const hipStream_t sws = (two && (idx & 1)) ? g_tws.aux : st;
        if (two && idx >= 2) HIP_OR_RET(hipStreamWaitEvent(sws, g_tws.walk_done[idx & 1], 0));    // this buffer's previous walk is done
        const int rc = run_chunk(c0, n_k, tb, cbytes, sws, idx & 1, by_rounds && two && n_k < chunk);
        if (rc) return rc;
    }
    if (two) HIP_OR_RET(hipStreamWaitEvent(st, g_tws.walk_done[(idx - 1) & 1], 0));      // (the walk stream is in order: the last walk covers all, the remainder's too)
    static thread_local char name[96];
    snprintf(name, sizeof name, "pmx_nwsg16q_kernel<%d,%d>/shared profile/packed trace + pmx_walkp_kernel/stats", G, R);
    g_last_kernel = name;
    return 0;
}

// pmx_long32_kernel over a batch, in chunks of bounded scratch: 0 done, 1 not eligible, < 0 error.  Score and end positions, 32-bit
// lanes, any gap model (open < extend included), alphabets up to 64 letters, no limit on either length.  Widths: local -- any
// (saturation = a score beyond the width); global / semi-global -- sat, 32, 64, or a fixed width whose range the boundary row /
// column already leaves (one pair: its lengths are known here); a fixed width that needs the range of H tracked is not served.
static int long_batch(const pmx_config_t *cfg, const DevMat &dm, const PmxBatch &b0, int64_t n, int32_t max_qlen, int32_t max_rlen,
                      pmx_record_t *d_out, hipStream_t st)
{
    if (pmx_env("PMX_NO_LONG_KERNEL") || cfg->matrix->type != PARASAIL_MATRIX_TYPE_SQUARE) return 1;
    int force_sat = 0, sat_above = 2147483647;
    const int wmax = cfg->width == 8 ? 127 : cfg->width == 16 ? 32767 : 2147483647;
    if (cfg->mode == PMX_MODE_SW) sat_above = wmax;
    else if (cfg->width == 8 || cfg->width == 16) {
        const bool pen_col = cfg->mode == PMX_MODE_NW || !(cfg->sg_flags & PMX_SG_QB), pen_row = cfg->mode == PMX_MODE_NW || !(cfg->sg_flags & PMX_SG_DB);
        const long long lo = std::min(pen_col ? -((long long)cfg->open + (long long)(max_qlen - 1) * cfg->extend) : 0LL,
                                      pen_row ? -((long long)cfg->open + (long long)(max_rlen - 1) * cfg->extend) : 0LL);
        if (n == 1 && lo < -(long long)wmax - 1) force_sat = 1; else return 1;     // (inside the range: the general kernel tracks min / max H)
    }
    // R = 4 (256-row bands) also for batches that fill the chip: 1 024-row bands (R = 16) share a step's fixed work among four
    // times the rows, but the longer dependent chain per step costs more (512 x 5 kbp^2: 5.0 ms against 6.5 ms, measured)
    int R = 4; long long bstride = 0; int nbmax = 0;
    // The form with two columns per step (pmx_long32_kernel_c2) has the shorter time per column, the one-column form the shorter
    // lag from band to band, and 128-row bands (R = 2) halve a step at twice the bands.  One call of a few pairs is a latency
    // problem: time = columns x (ns per column) + bands x (ns of lag per band), constants measured on MI355X per form
    // (profiles/r04/long_shapes.txt); a batch that fills the chip keeps the one-column form with 256-row bands (throughput, measured).
    int two_cols = 0;
    if (n <= 16) {
        const bool sw = cfg->mode == PMX_MODE_SW;
        const double nb4 = (max_qlen + 255) / 256, nb2 = (max_qlen + 127) / 128, cols = max_rlen;
        // (ns per column and ns per band, measured at the end of round 4: local alignment in the plain form, global / semi-global in the
        //  form with column skew and row offset -- profiles/r04/long_shapes.txt, long_single_forms.txt)
        const double t1 = cols * (sw ? 156 : 115) + nb4 * (sw ? 18400 : 15700);
        const double t24 = cols * (sw ? 111 : 78.6) + nb4 * (sw ? 32600 : 20100);
        const double t22 = cols * (sw ? 85 : 62.4) + nb2 * (sw ? 23100 : 16400);
        if (t22 < 0.95 * t1 && t22 <= t24) { two_cols = 1; R = 2; }          // (within 5 %: the first form)
        else if (t24 < 0.95 * t1) two_cols = 1;
    }
    if (pmx_env("PMX_LONG_TWO_COLUMNS")) two_cols = 1;
    if (pmx_env("PMX_LONG_ONE_COLUMN")) { two_cols = 0; R = 4; }
    if (const char *e = pmx_env("PMX_LONG_ROWS_PER_LANE")) R = atoi(e) == 2 ? 2 : atoi(e) == 16 ? 16 : 4;
    if (R == 16) two_cols = 0;
    size_t per_pair = pmx_long_scratch_bytes(1, max_qlen, max_rlen, R, &bstride, &nbmax);
    if (per_pair > ((size_t)4 << 30)) { R = 16; two_cols = 0; per_pair = pmx_long_scratch_bytes(1, max_qlen, max_rlen, R, &bstride, &nbmax); }
    size_t fb = 0, tb = 0;
    if (hipMemGetInfo(&fb, &tb) != hipSuccess) fb = 0;
    size_t budget = std::min<size_t>((size_t)4 << 30, fb / 4) + ((size_t)64 << 20);
    if (const char *e = pmx_env("PMX_LONG_CHUNK_BYTES")) budget = (size_t)atof(e);        // tests force several chunks
    if (per_pair > budget && !pmx_env("PMX_LONG_CHUNK_BYTES")) return 1;
    int64_t chunk = (int64_t)(budget / per_pair);
    if (chunk < 1) chunk = 1;
    if (chunk > n) chunk = n;
    void *scr = nullptr;
    if (scratch_reserve((size_t)chunk * per_pair, &scr, SCR_LONG)) return 1;
    // the bands of a pair wait for one another across workgroups; the wait is bounded (pmx_long.hip): ~2 us a poll
    int spin_limit = 1 << 20;
    if (const char *e = pmx_env("PMX_LONG_SPIN_LIMIT")) spin_limit = atoi(e);             // tests force the give-up path
    int chunk_cols = 16;
    if (const char *e = pmx_env("PMX_LONG_CHUNK_COLS")) chunk_cols = atoi(e) == 64 ? 64 : 16;
    HIP_OR_RET(hipMemsetAsync(scr, 0, 64, st));
    for (int64_t c0 = 0; c0 < n; c0 += chunk) {
        PmxBatch b = b0;
        b.perm = nullptr;                                  // (a processing order is a hint: records are indexed by pair)
        b.n = (n - c0 < chunk) ? n - c0 : chunk;
        if (!b.q_shared) b.qoff = b0.qoff + c0;
        b.roff = b0.roff + c0;
        const int rc = pmx_launch_long(b, dm.d, cfg->mode, cfg->sg_flags, cfg->open, cfg->extend, R, scr, d_out + c0, sat_above, force_sat, st, spin_limit, chunk_cols, two_cols);
        if (rc < 0) { set_err("long-pair kernel launch failed: %s", hipGetErrorString((hipError_t)(-rc))); return rc; }
        if (rc) return c0 == 0 ? 1 : (set_err("long-pair kernel refused a later chunk"), -1);
    }
    // did a band give up waiting?  (The one host synchronisation of this path; one-pair calls synchronise right after anyway.)
    // (into pinned memory: an asynchronous copy to pageable memory goes through the runtime's staging thread, and the synchronisation
    //  behind it was seen to take 20-30 ms in steps of 10 ms for a 5 ms batch)
    static thread_local int *pin_flag = nullptr;
    if (!pin_flag) HIP_OR_RET(hipHostMalloc((void **)&pin_flag, 64, hipHostMallocDefault));
    *pin_flag = 0;
    HIP_OR_RET(hipMemcpyAsync(pin_flag, scr, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_OR_RET(hipStreamSynchronize(st));
    const int gave_up = *pin_flag;
    if (gave_up) {
        set_err("long-pair kernel: a band's bounded wait for the band above ran out (dispatch order assumption broken, or PMX_LONG_SPIN_LIMIT); "
                "the call was redone on the per-pair kernels");
        return 1;
    }
    g_last_kernel = R == 16 ? "pmx_long32_kernel<16>/bands across the chip"
                  : two_cols ? (R == 2 ? "pmx_long32_kernel_c2<2>/bands across the chip, two columns per step" : "pmx_long32_kernel_c2<4>/bands across the chip, two columns per step")
                  : (R == 2 ? "pmx_long32_kernel<2>/bands across the chip" : "pmx_long32_kernel<4>/bands across the chip");
    return 0;
}

// Device-resident batch.  q_shared > 0: every pair uses the one query d_qbuf[0..q_shared) (profile arm).
static int run_batch_device(const pmx_config_t *cfg, int64_t n,
                            const uint8_t *d_qbuf, const int64_t *d_qoff, int q_shared,
                            const uint8_t *d_rbuf, const int64_t *d_roff,
                            int32_t max_qlen, int32_t max_rlen,
                            pmx_record_t *d_out, pmx_stats_t *d_stats_out, void *stream,
                            int q_shared_wild /* shared query: it holds a letter beyond the first four (or unknown) */)
{
    if (check_cfg(cfg)) return -1;
    if (n <= 0) return 0;
    if (max_qlen <= 0 || max_rlen <= 0) { set_err("max_qlen / max_rlen must be positive"); return -1; }
    if ((cfg->want & PMX_WANT_STATS) && !d_stats_out) { set_err("stats requested without a stats buffer"); return -1; }
    if (cfg->want & PMX_WANT_CIGAR) { set_err("use pmx_align_batch_cigar for CIGAR output"); return -1; }
    DevMat dm;
    if (get_devmat(cfg->matrix, &dm)) return -1;
    hipStream_t st = (hipStream_t)stream;
    PmxBatch b = {d_qbuf, d_qoff, d_rbuf, d_roff, n, max_qlen, max_rlen, q_shared, nullptr, nullptr, nullptr, 0, 0};
    const int want = cfg->want & ~PMX_WANT_SORTED;
    if ((cfg->want & PMX_WANT_SORTED) && n >= 64 && n < (1LL << 32)) {
        void *scr = nullptr;
        if (scratch_reserve(pmx_sort_scratch_bytes(n), &scr, SCR_SORT)) return -1;
        const int rc = pmx_build_length_perm(d_roff, n, scr, &b.perm, st);
        if (rc < 0) { set_err("length sort failed (%d)", rc); return rc; }
    }
    // Few long pairs (one align() call on kilobases: src/aligner/mod.rs:397-430 has no length limit), or queries beyond the packed
    // kernels' 2 048 rows in any number: the query's bands spread over the chip (pmx_long.hip).
    long long long_min_cells = 250000;                     // (600 x 600: 0.18 ms here, 0.23-0.27 ms in the one-wave packed kernels)
    if (const char *e = pmx_env("PMX_LONG_MIN_CELLS")) long_min_cells = atoll(e);
    if (want == 0 && ((n <= 16 && max_qlen >= 512 && (long long)max_qlen * max_rlen >= long_min_cells) || max_qlen > 2048)) {
        const int rc = long_batch(cfg, dm, b, n, max_qlen, max_rlen, d_out, st);
        if (rc <= 0) return rc;
    }
    if (fast_sw_eligible(cfg)) {
        b.q_has_wildcard = q_shared ? q_shared_wild : 0;
        b.sat_above = cfg->width == 8 ? 127 : 0;
        if (n >= 4096 && n < (1LL << 32) && !b.q_has_wildcard) {
            // scratch that lets the launcher pick a kernel which hands some pairs back for a second launch
            void *scr = nullptr;
            if (scratch_reserve(((size_t)n + 1) * sizeof(unsigned), &scr, SCR_RETRY)) return -1;
            b.retry_count = (int *)scr;
            b.retry_list = (unsigned *)scr + 1;
        }
        const int rc = pmx_launch_sw16(b, dm.d, cfg->open, cfg->extend, d_out, st, &g_last_kernel);
        if (rc < 0) { set_err("sw16 launch failed: %s", hipGetErrorString((hipError_t)(-rc))); return rc; }
        if (rc == 0) {
            // Overflow promotion (`sat`, 32, 64): pairs whose int16 lanes overflowed are re-run in the
            // 32-bit kernel.  Skipped without any synchronisation when no score can reach 32768.
            const long long bound_score = (long long)(max_qlen < max_rlen ? max_qlen : max_rlen) *
                                          (cfg->matrix->max > 0 ? cfg->matrix->max : 0);
            // (the max3 variant of the fast kernel is exact up to 29 696 - max score; beyond that it sets
            //  PMX_FLAG_RERUN and the pair is redone here whatever the requested width)
            if (bound_score <= 27000) return 0;
            const int mask = PMX_FLAG_RERUN | ((cfg->width == 16 || cfg->width == 8) ? 0 : PMX_FLAG_SATURATED);
            DevBuf<int64_t> list; DevBuf<int> cnt;
            if (list.try_alloc((size_t)n) || cnt.try_alloc(1)) { set_err("out of device memory (promotion list)"); return -2; }
            int rc2 = pmx_launch_collect_saturated(d_out, n, list.p, cnt.p, mask, st);
            if (rc2) { set_err("collect kernel failed (%d)", rc2); return rc2; }
            int count = 0;
            HIP_OR_RET(hipMemcpyAsync(&count, cnt.p, sizeof(int), hipMemcpyDeviceToHost, st));
            HIP_OR_RET(hipStreamSynchronize(st));
            if (count == 0) return 0;
            DevBuf<int32_t> bnd; DevBuf<uint8_t> rsb;
            const size_t stride2 = (size_t)8 * max_rlen;
            const bool fits2 = pmx_general_lds_fits(dm.d.msize, dm.d.msize, max_rlen);
            const size_t rs2 = fits2 ? 0 : (((size_t)max_rlen + 8 + 15) & ~(size_t)15);
            if (bnd.try_alloc((size_t)count * stride2) || (!fits2 && rsb.try_alloc((size_t)count * rs2))) {
                set_err("out of device memory (promotion pass of %d pairs)", count); return -2;
            }
            PmxGeneralArgs a; memset(&a, 0, sizeof a);
            a.qbuf = d_qbuf; a.qoff = q_shared ? nullptr : d_qoff; a.shared_qlen = q_shared;
            a.rbuf = d_rbuf; a.roff = d_roff; a.n = count; a.index = list.p; a.max_rlen = max_rlen;
            a.scores = dm.d.scores; a.mapper = dm.d.mapper; a.msize = dm.d.msize; a.mat_rows = dm.d.msize;
            a.mode = cfg->mode; a.sg_flags = cfg->sg_flags; a.open = cfg->open; a.ext = cfg->extend; a.band = -1;
            a.bits = cfg->width == 16 ? 16 : cfg->width == 8 ? 8 : 32; a.bound = bnd.p; a.bound_stride = (long long)stride2; a.rec = d_out;
            if (!fits2) { a.rs_scratch = rsb.p; a.rs_stride = (long long)rs2; }
            rc2 = pmx_launch_general(a, false, st);
            if (rc2) { set_err("promotion launch failed (%d)", rc2); return rc2 < 0 ? rc2 : -1; }
            HIP_OR_RET(hipStreamSynchronize(st));      // scratch is released on return
            return 0;
        }
        // rc == 1: shape not covered by the fast kernel -> general kernel below
    }
    // statistics: (0) small alphabets in full batches: counts along the packed traceback; (1) the packed statistics kernel
    // (shared profile, or per-pair over a large alphabet, no free end); (2) large alphabets with short references: traceback
    // again; (3) the unpacked statistics kernel
    if (q_shared && want == PMX_WANT_STATS && cfg->width != 8 && cfg->matrix->type == PARASAIL_MATRIX_TYPE_SQUARE &&
        (cfg->mode == PMX_MODE_NW || cfg->mode == PMX_MODE_SG) && (n >= 512 || pmx_env("PMX_STATS_BY_TRACE"))) {
        // profile arm with statistics (BASELINE config 3): traceback sweep + counting walk
        const int rc = stats_by_trace_shared(cfg, dm, b, d_out, d_stats_out, st);
        if (rc < 0) return rc;
        if (rc == 0) return 0;
    }
    if (upload_wait(INT64_MAX, st, 0)) return -1;             // (host entry with a sliced upload: every other path reads the whole batch)
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            if (want == PMX_WANT_STATS && cfg->width != 8 && cfg->matrix->type == PARASAIL_MATRIX_TYPE_SQUARE) {
                // second generation (two pairs per lane slot); global / semi-global inside its exact window
                const int rc = pmx_launch_stats16p(b, dm.d, cfg->mode, cfg->sg_flags, cfg->open, cfg->extend, d_out, d_stats_out, st, &g_last_kernel);
                if (rc < 0) { set_err("stats16p launch failed: %s", hipGetErrorString((hipError_t)(-rc))); return rc; }
                if (rc == 0) return 0;
            }
            if (!(dm.d.msize > 8 && (max_rlen <= 1024 || pmx_env("PMX_STATS_BY_TRACE_ANY")))) break;
        }
        if (want == PMX_WANT_STATS && cfg->width != 8 && cfg->matrix->type == PARASAIL_MATRIX_TYPE_SQUARE &&
            (dm.d.msize <= 8 || pass == 1) && !q_shared && (n >= 2048 || pmx_env("PMX_STATS_BY_TRACE")) && !pmx_env("PMX_NO_STATS_BY_TRACE")) {
            // (a few pairs: the one-pass statistics kernel has the lower latency)
            // Small alphabets: statistics = counts along the traceback path (the same decisions and tie-breaks as the
            // coupled statistics tables).  The packed traceback sweep runs at more than twice the speed of the
            // statistics kernel and the walk is cheap; the trace scratch is bounded by working in chunks (same stream,
            // no host synchronisation).  Large alphabets take this route for short references only (measured: per-pair
            // protein 285 x 285, sw 0.37 -> 1.03 TCUPS, nw 0.40 -> 0.89 with the matrix-lookup traceback kernels; against 5-kaa references the staged references
            // and per-pair profiles starve the 16-rows-per-lane traceback shapes and the statistics kernel wins).
            PmxBatch bt = b; bt.perm = nullptr;
            int variant = 0, Tmax = 0; size_t tbytes = 0;
            if (pmx_trace16_plan(bt, dm.d, cfg->mode, cfg->open, cfg->extend, &variant, &Tmax, &tbytes) == 0 && variant >= 10) {
                double budget = 8e9;
                { size_t fb = 0, tb = 0; if (hipMemGetInfo(&fb, &tb) == hipSuccess && 0.15 * (double)fb < budget) budget = 0.15 * (double)fb; }
                const double per_pair = (double)tbytes / (double)n + 1.0;
                int64_t nchunks = (int64_t)((double)tbytes / budget) + 1;
                if (nchunks < 2 && n >= 16384) nchunks = 2;            // two chunks at least: the walk of one runs beside the sweep of the next
                int64_t chunk = ((n + nchunks - 1) / nchunks + 63) / 64 * 64;
                if ((double)chunk * per_pair > budget) chunk = (int64_t)(budget / per_pair) / 64 * 64;
                if (chunk < 64) chunk = 64;
                if (chunk > n) chunk = n;
                bt.n = chunk;
                (void)pmx_trace16_plan(bt, dm.d, cfg->mode, cfg->open, cfg->extend, &variant, &Tmax, &tbytes);
                const size_t cbytes = (tbytes + 255) & ~(size_t)255;
                const bool two = chunk < n;
                if (two && trace_ws_init()) return -1;
                uint32_t *tbuf = nullptr;
                if (scratch_reserve(cbytes * (two ? 2 : 1), (void **)&tbuf, SCR_TRACE)) return -1;
                const size_t fstride = (size_t)chunk / 2 + 16;           // per-block flags of a chunk's sweep, two sets (as in cigar_device_run)
                int *bflags = nullptr;
                if (scratch_reserve(2 * fstride * sizeof(int), (void **)&bflags, SCR_RETRY)) return -1;
                // as in the batch CIGAR entry: two trace buffers, the counting walk of chunk c on the walk stream beside the sweep of
                // chunk c + 1, sweeps alternating between the caller's stream and an internal one
                if (two) { HIP_OR_RET(hipEventRecord(g_tws.start, st)); HIP_OR_RET(hipStreamWaitEvent(g_tws.aux, g_tws.start, 0)); }
                int idx = 0;
                for (int64_t c0 = 0; c0 < n; c0 += chunk, ++idx) {
                    PmxBatch bc = bt;
                    bc.n = (n - c0 < chunk) ? n - c0 : chunk;
                    bc.qoff = d_qoff + c0; bc.roff = d_roff + c0;
                    bc.blockflag = bflags + (size_t)(idx & 1) * fstride;
                    const hipStream_t sws = (two && (idx & 1)) ? g_tws.aux : st;
                    if (two && idx >= 2) HIP_OR_RET(hipStreamWaitEvent(sws, g_tws.walk_done[idx & 1], 0));
                    PmxWalkSplit sp = {two ? g_tws.walk : st, g_tws.sweep_done[idx & 1], two ? g_tws.walk_done[idx & 1] : nullptr, 0, nullptr};
                    const int rc = pmx_launch_trace16(variant, bc, dm.d, cfg->mode, cfg->sg_flags, cfg->open, cfg->extend, d_out + c0,
                                                      (uint32_t *)((unsigned char *)tbuf + (two ? (size_t)(idx & 1) * cbytes : 0)), Tmax,
                                                      nullptr, nullptr, nullptr, nullptr, sws, d_stats_out + c0, two ? &sp : nullptr);
                    if (rc) { set_err("stats-by-traceback launch failed (%d)", rc); return rc < 0 ? rc : -1; }
                }
                if (two) HIP_OR_RET(hipStreamWaitEvent(st, g_tws.walk_done[(idx - 1) & 1], 0));
                g_last_kernel = variant >= 20 ? "pmx_sw16_kernel/packed trace + pmx_walkp_kernel/stats" : "pmx_nwsg16v_kernel/packed trace + pmx_walkp_kernel/stats";
                return 0;
            }
        }
    }
    if (want == PMX_WANT_STATS && cfg->width != 8 && cfg->matrix->type == PARASAIL_MATRIX_TYPE_SQUARE) {
        const int rc = pmx_launch_stats16(b, dm.d, cfg->mode, cfg->sg_flags, cfg->open, cfg->extend, d_out, d_stats_out, st, &g_last_kernel);
        if (rc < 0) { set_err("stats16 launch failed: %s", hipGetErrorString((hipError_t)(-rc))); return rc; }
        if (rc == 0) return 0;
    }
    if ((cfg->mode == PMX_MODE_NW || cfg->mode == PMX_MODE_SG) && want == 0 && (cfg->width != 8 || !pmx_env("PMX_NWSG8_GENERAL")) &&
        cfg->matrix->type == PARASAIL_MATRIX_TYPE_SQUARE) {
        // (width 8: the same int16 kernels, which then also track the range of H for the saturation flag -- the reference's
        //  narrowest width is its fastest on a CPU; it must not be the slow road here)
        b.track8 = cfg->width == 8;
        if (dm.d.msize <= 5 && n >= 2048 && !q_shared) {       // per-block flags: lets the launcher try the perm-table form first
            void *scr = nullptr;
            if (scratch_reserve(((size_t)n / 2 + 16) * sizeof(int), &scr, SCR_RETRY)) return -1;
            b.blockflag = (int *)scr;
        }
        const int rc = pmx_launch_nwsg16(b, dm.d, cfg->mode, cfg->sg_flags, cfg->open, cfg->extend, d_out, st, &g_last_kernel);
        if (rc < 0) { set_err("nwsg16 launch failed: %s", hipGetErrorString((hipError_t)(-rc))); return rc; }
        if (rc == 0) return 0;     // the host-side range proof makes overflow impossible: no promotion pass
    }
    if (cfg->matrix->type == PARASAIL_MATRIX_TYPE_PSSM) { set_err("PSSM matrices are single-pair only"); return -1; }
    if (want == 0) {
        // Outside every packed kernel's window (gap models with open < extend, alphabets of 32 and more letters, value ranges the
        // int16 lanes cannot prove): the 32-bit band kernel, one wave per 256 query rows -- several times the general kernel's rate
        const int rc = long_batch(cfg, dm, b, n, max_qlen, max_rlen, d_out, st);
        if (rc <= 0) return rc;
    }
    const int rcg = general_batch(cfg, dm, n, d_qbuf, d_qoff, q_shared, d_rbuf, d_roff, max_rlen, -1, nullptr,
                                  (want & PMX_WANT_STATS) != 0, d_out, d_stats_out, st, max_qlen);
    if (rcg) return rcg;
    g_last_kernel = "pmx_general_kernel";
    return 0;
}

// The device entries keep internal scratch (length-sort permutation, retry list, trace records, op slots) per HOST THREAD.  A thread
// that queues its next call on ANOTHER stream would let that call overwrite scratch the previous call's kernels may still read:
// every device entry therefore ends by recording an event on its stream, and a call that arrives on a different stream first makes
// its stream wait for that event.  Calls on one stream cost nothing extra; calls from different threads never share scratch.
struct StreamGuard {
    static thread_local hipEvent_t ev; static thread_local hipStream_t last; static thread_local int dev; static thread_local bool armed;
    hipStream_t st; bool ok;
    explicit StreamGuard(void *stream) : st((hipStream_t)stream), ok(true)
    {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess) { ok = false; return; }
        if (dev != d) { if (ev) (void)hipEventDestroy(ev); ev = nullptr; armed = false; dev = d; }
        if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { ok = false; return; }
        if (armed && last != st && hipStreamWaitEvent(st, ev, 0) != hipSuccess) ok = false;
    }
    ~StreamGuard() { if (ok && ev && hipEventRecord(ev, st) == hipSuccess) { last = st; armed = true; } }
};
thread_local hipEvent_t StreamGuard::ev = nullptr;
thread_local hipStream_t StreamGuard::last = nullptr;
thread_local int StreamGuard::dev = -1;
thread_local bool StreamGuard::armed = false;

extern "C" int pmx_align_batch_device(const pmx_config_t *cfg, int64_t n,
                                      const uint8_t *d_qbuf, const int64_t *d_qoff,
                                      const uint8_t *d_rbuf, const int64_t *d_roff,
                                      int32_t max_qlen, int32_t max_rlen,
                                      pmx_record_t *d_out, pmx_stats_t *d_stats_out, void *stream)
{
    StreamGuard guard(stream);
    if (!guard.ok) { set_err("stream guard failed"); return -1; }
    return run_batch_device(cfg, n, d_qbuf, d_qoff, 0, d_rbuf, d_roff, max_qlen, max_rlen, d_out, d_stats_out, stream);
}

static void host_maxlens(int64_t n, const int64_t *off, int32_t *mx, bool *bad, int32_t *mn = nullptr)
{
    int64_t m = 0, lo = INT32_MAX;                  // (no stores inside the loop: one compare-select pair per element)
    for (int64_t k = 0; k < n; ++k) {
        const int64_t l = off[k + 1] - off[k];
        m = l > m ? l : m;
        lo = l < lo ? l : lo;
    }
    if (lo <= 0 || m > INT32_MAX) *bad = true;
    *mx = (int32_t)(m > INT32_MAX ? INT32_MAX : m);
    if (mn) *mn = (int32_t)(lo < 0 ? 0 : lo);
}
// The same over both offset arrays of a large batch, split over a few host threads: the scan of 2 x 1M offsets is 1.2 ms on one
// core, as long as a third of the device work it precedes.
struct LenScan { int32_t mq = 0, mr = 0, mnr = INT32_MAX; bool bad = false; };
static LenScan scan_lengths(int64_t n, const int64_t *qoff, const int64_t *roff)
{
    const int T = n >= 262144 ? 4 : 1;
    LenScan part[4];
    auto work = [&](int t) {
        const int64_t a = n * t / T, e = n * (t + 1) / T;
        host_maxlens(e - a, qoff + a, &part[t].mq, &part[t].bad);
        host_maxlens(e - a, roff + a, &part[t].mr, &part[t].bad, &part[t].mnr);
    };
    // (thread creation can fail -- a process at its thread limit: std::system_error must not unwind through the C ABI; the
    //  parts without a helper are scanned here)
    std::thread th[3];
    bool started[3] = {false, false, false};
    for (int t = 1; t < T; ++t) {
        try { th[t - 1] = std::thread(work, t); started[t - 1] = true; } catch (const std::system_error &) {}
    }
    work(0);
    for (int t = 1; t < T; ++t) { if (started[t - 1]) th[t - 1].join(); else work(t); }
    LenScan r = part[0];
    for (int t = 1; t < T; ++t) {
        r.mq = std::max(r.mq, part[t].mq); r.mr = std::max(r.mr, part[t].mr); r.mnr = std::min(r.mnr, part[t].mnr); r.bad |= part[t].bad;
    }
    return r;
}
// ragged reference lengths: worth a length-sorted processing order
static pmx_config_t with_sort_hint(const pmx_config_t *cfg, int32_t min_rlen, int32_t max_rlen, int64_t n)
{
    pmx_config_t c = *cfg;
    if (n >= 256 && (long long)min_rlen * 5 < (long long)max_rlen * 4) c.want |= PMX_WANT_SORTED;
    return c;
}

// Host buffers in, host records out.  packed2: the sequence buffers hold 2 bits per base (base b in byte b / 4 at bits 2 (b % 4),
// code c = letter c of the matrix alphabet) and the offsets count bases: a quarter of the bytes cross PCIe and a small kernel
// spells them out into the staging buffers before the slice is aligned.
static int host_batch(const pmx_config_t *cfg, int64_t n,
                      const uint8_t *qbuf, const int64_t *qoff, const uint8_t *rbuf, const int64_t *roff,
                      pmx_record_t *out, pmx_stats_t *stats_out, bool packed2)
{
    if (check_cfg(cfg)) return -1;
    if (n <= 0) return 0;
    if (!qbuf || !qoff || !rbuf || !roff || !out) { set_err("null buffer"); return -1; }
    // the length scan runs beside the first transfers (it needs the host only; the offsets go up meanwhile)
    LenScan ls;
    std::future<void> scan;
    if (n >= 262144) {
        try { scan = std::async(std::launch::async, [&]() { ls = scan_lengths(n, qoff, roff); }); }
        catch (const std::system_error &) { ls = scan_lengths(n, qoff, roff); }          // no helper thread to be had: scan inline
    } else ls = scan_lengths(n, qoff, roff);
    struct ScanJoin { std::future<void> &f; ~ScanJoin() { if (f.valid()) f.wait(); } } scan_join{scan};     // (early returns)
    if (qoff[0] != 0 || roff[0] != 0) { set_err("offset arrays must start at 0"); return -1; }
    if (qoff[n] <= 0 || roff[n] <= 0 || qoff[n] > ((int64_t)1 << 40) || roff[n] > ((int64_t)1 << 40)) { set_err("bad offset arrays"); return -1; }
    const size_t qbytes = (size_t)qoff[n], rbytes = (size_t)roff[n];
    uint32_t letters = 0;
    if (packed2) {
        const char *al = cfg->matrix->alphabet;
        if (!al || strlen(al) < 4) { set_err("2-bit input needs a matrix alphabet of at least four letters"); return -1; }
        letters = (uint32_t)(unsigned char)al[0] | ((uint32_t)(unsigned char)al[1] << 8) | ((uint32_t)(unsigned char)al[2] << 16) | ((uint32_t)(unsigned char)al[3] << 24);
    }
    // device staging is kept per host thread between calls (hipMalloc / hipFree of hundreds of MB cost milliseconds)
    struct { uint8_t *p; } dq, dr, dq2 = {nullptr}, dr2 = {nullptr}; struct { int64_t *p; } dqo, dro; struct { pmx_record_t *p; } drec; struct { pmx_stats_t *p; } dst = {nullptr};
    const bool stats = cfg->want & PMX_WANT_STATS;
    if (stats && !stats_out) { set_err("stats requested without a stats buffer"); return -1; }
    if (scratch_reserve(qbytes + 16, (void **)&dq.p, SCR_HQ) || scratch_reserve(rbytes + 16, (void **)&dr.p, SCR_HR) ||
        scratch_reserve(sizeof(int64_t) * (n + 1), (void **)&dqo.p, SCR_HQO) || scratch_reserve(sizeof(int64_t) * (n + 1), (void **)&dro.p, SCR_HRO) ||
        scratch_reserve(sizeof(pmx_record_t) * n, (void **)&drec.p, SCR_HREC) ||
        (stats && scratch_reserve(sizeof(pmx_stats_t) * n, (void **)&dst.p, SCR_HST)) ||
        (packed2 && (scratch_reserve(qbytes / 4 + 16, (void **)&dq2.p, SCR_HQ2) || scratch_reserve(rbytes / 4 + 16, (void **)&dr2.p, SCR_HR2)))) return -1;
    static thread_local hipStream_t s_copy = nullptr, s_comp = nullptr, s_back = nullptr;
    static thread_local hipEvent_t s_ev[8], s_done[8];
    static thread_local int s_dev = -1;
    int dev = 0; HIP_OR_RET(hipGetDevice(&dev));
    if (s_dev != dev) {
        if (s_copy) {                                      // the thread moved to another device: release the old device's objects
            (void)hipStreamDestroy(s_copy); (void)hipStreamDestroy(s_comp); (void)hipStreamDestroy(s_back);
            for (auto &e : s_ev) (void)hipEventDestroy(e);
            for (auto &e : s_done) (void)hipEventDestroy(e);
        }
        HIP_OR_RET(hipStreamCreateWithFlags(&s_copy, hipStreamNonBlocking));
        HIP_OR_RET(hipStreamCreateWithFlags(&s_comp, hipStreamNonBlocking));
        HIP_OR_RET(hipStreamCreateWithFlags(&s_back, hipStreamNonBlocking));
        for (auto &e : s_ev) HIP_OR_RET(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &e : s_done) HIP_OR_RET(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        s_dev = dev;
    }
    HIP_OR_RET(hipMemcpyAsync(dqo.p, qoff, sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, s_copy));
    HIP_OR_RET(hipMemcpyAsync(dro.p, roff, sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, s_copy));
    if (scan.valid()) scan.get();
    if (ls.bad) { (void)hipStreamSynchronize(s_copy); set_err("every sequence must have length >= 1"); return -1; }
    const int32_t mq = ls.mq, mr = ls.mr;
    const pmx_config_t cfg_s = with_sort_hint(cfg, ls.mnr, mr, n);
    cfg = &cfg_s;
    // Large uniform batches: the sequence bytes go up in slices on a copy stream while the previous slice is already being aligned
    // on a compute stream (the offsets are absolute, so a slice is just a pointer shift) and the slice before that travels back on a
    // third; over PCIe the transfer is several times the kernel time, this hides the kernel and the return trip behind it.
    const int K = (n >= 262144 && !(cfg->want & PMX_WANT_SORTED)) ? (packed2 ? 4 : 8) : 1;     // (2-bit input: the kernel, not the link, is the longer leg)
    for (int sl = 0; sl < K; ++sl) {
        const int64_t a = n * sl / K, e = n * (sl + 1) / K;
        if (e <= a) continue;
        if (packed2) {
            const int64_t qa = qoff[a] / 4, qe = (qoff[e] + 3) / 4, ra = roff[a] / 4, re = (roff[e] + 3) / 4;
            HIP_OR_RET(hipMemcpyAsync(dq2.p + qa, qbuf + qa, (size_t)(qe - qa), hipMemcpyHostToDevice, s_copy));
            HIP_OR_RET(hipMemcpyAsync(dr2.p + ra, rbuf + ra, (size_t)(re - ra), hipMemcpyHostToDevice, s_copy));
        } else {
            HIP_OR_RET(hipMemcpyAsync(dq.p + qoff[a], qbuf + qoff[a], (size_t)(qoff[e] - qoff[a]), hipMemcpyHostToDevice, s_copy));
            HIP_OR_RET(hipMemcpyAsync(dr.p + roff[a], rbuf + roff[a], (size_t)(roff[e] - roff[a]), hipMemcpyHostToDevice, s_copy));
        }
        HIP_OR_RET(hipEventRecord(s_ev[sl], s_copy));
        HIP_OR_RET(hipStreamWaitEvent(s_comp, s_ev[sl], 0));
        if (packed2) {
            int rc2 = pmx_launch_unpack2(dq2.p, dq.p, qoff[a], qoff[e], letters, s_comp);
            if (!rc2) rc2 = pmx_launch_unpack2(dr2.p, dr.p, roff[a], roff[e], letters, s_comp);
            if (rc2) { (void)hipStreamSynchronize(s_comp); set_err("2-bit unpack launch failed (%d)", rc2); return rc2; }
        }
        const int rc = pmx_align_batch_device(cfg, e - a, dq.p, dqo.p + a, dr.p, dro.p + a, mq, mr, drec.p + a,
                                              stats ? dst.p + a : nullptr, s_comp);
        if (rc) { (void)hipStreamSynchronize(s_comp); return rc; }
        HIP_OR_RET(hipEventRecord(s_done[sl], s_comp));
        }
    // the way back, slice by slice as they finish (a copy into pageable host memory blocks the host thread, so it is not issued
    // inside the loop above: the later slices are already queued and keep the GPU busy meanwhile)
    for (int sl = 0; sl < K; ++sl) {
        const int64_t a = n * sl / K, e = n * (sl + 1) / K;
        if (e <= a) continue;
        HIP_OR_RET(hipEventSynchronize(s_done[sl]));
            HIP_OR_RET(hipMemcpyAsync(out + a, drec.p + a, sizeof(pmx_record_t) * (size_t)(e - a), hipMemcpyDeviceToHost, s_back));
        if (stats) HIP_OR_RET(hipMemcpyAsync(stats_out + a, dst.p + a, sizeof(pmx_stats_t) * (size_t)(e - a), hipMemcpyDeviceToHost, s_back));
    }
    HIP_OR_RET(hipStreamSynchronize(s_back));
    return 0;
}

extern "C" int pmx_align_batch(const pmx_config_t *cfg, int64_t n,
                               const uint8_t *qbuf, const int64_t *qoff,
                               const uint8_t *rbuf, const int64_t *roff,
                               pmx_record_t *out, pmx_stats_t *stats_out)
{
    return host_batch(cfg, n, qbuf, qoff, rbuf, roff, out, stats_out, false);
}

extern "C" int pmx_align_batch_2bit(const pmx_config_t *cfg, int64_t n,
                                    const uint8_t *q2, const int64_t *qoff,
                                    const uint8_t *r2, const int64_t *roff,
                                    pmx_record_t *out, pmx_stats_t *stats_out)
{
    return host_batch(cfg, n, q2, qoff, r2, roff, out, stats_out, true);
}

extern "C" int pmx_align_profile_batch(const pmx_config_t *cfg, const parasail_profile_t *profile, int64_t n,
                                       const uint8_t *rbuf, const int64_t *roff,
                                       pmx_record_t *out, pmx_stats_t *stats_out)
{
    if (check_cfg(cfg)) return -1;
    if (!profile) { set_err("null profile"); return -1; }
    if (n <= 0) return 0;
    if (!rbuf || !roff || !out) { set_err("null buffer"); return -1; }
    if (profile->matrix != cfg->matrix) { set_err("profile was built with a different matrix"); return -1; }
    if (cfg->matrix->type == PARASAIL_MATRIX_TYPE_PSSM) { set_err("PSSM matrices are single-pair only"); return -1; }
    int32_t mr = 0, mnr = 0; bool bad = false;
    host_maxlens(n, roff, &mr, &bad, &mnr);
    if (bad || roff[0] != 0) { set_err("bad reference offsets"); return -1; }
    const pmx_config_t cfg_s = with_sort_hint(cfg, mnr, mr, n);
    cfg = &cfg_s;
    const bool stats = (cfg->want & PMX_WANT_STATS) != 0;
    if (stats && !stats_out) { set_err("stats requested without a stats buffer"); return -1; }
    struct { const uint8_t *p; } dq;
    if (profile_device_query(profile, &dq.p)) return -1;
    // References go up in slices on a copy stream while the previous slice is aligned (a slice is sorted and aligned on its own:
    // the offsets are absolute, a slice is a pointer shift) and finished slices travel back; the device staging is kept per host
    // thread between calls.  cfg 5's eighth (3.4 GB of references) spends 60 ms on the link, all of it behind the kernels.
    const size_t rbytes = (size_t)roff[n];
    struct { uint8_t *p; } dr; struct { int64_t *p; } dro; struct { pmx_record_t *p; } drec; struct { pmx_stats_t *p; } dst = {nullptr};
    if (scratch_reserve(rbytes + 16, (void **)&dr.p, SCR_HR) || scratch_reserve(sizeof(int64_t) * (n + 1), (void **)&dro.p, SCR_HRO) ||
        scratch_reserve(sizeof(pmx_record_t) * n, (void **)&drec.p, SCR_HREC) ||
        (stats && scratch_reserve(sizeof(pmx_stats_t) * n, (void **)&dst.p, SCR_HST))) return -1;
    static thread_local hipStream_t s_copy = nullptr, s_comp = nullptr;
    static thread_local hipEvent_t s_up[8], s_done[8];
    static thread_local int s_dev = -1;
    int dev = 0; HIP_OR_RET(hipGetDevice(&dev));
    if (s_dev != dev) {
        if (s_copy) { (void)hipStreamDestroy(s_copy); (void)hipStreamDestroy(s_comp); for (int k = 0; k < 8; ++k) { (void)hipEventDestroy(s_up[k]); (void)hipEventDestroy(s_done[k]); } }
        HIP_OR_RET(hipStreamCreateWithFlags(&s_copy, hipStreamNonBlocking));
        HIP_OR_RET(hipStreamCreateWithFlags(&s_comp, hipStreamNonBlocking));
        for (int k = 0; k < 8; ++k) {
            HIP_OR_RET(hipEventCreateWithFlags(&s_up[k], hipEventDisableTiming));
            HIP_OR_RET(hipEventCreateWithFlags(&s_done[k], hipEventDisableTiming));
        }
        s_dev = dev;
    }
    // (a slice must still fill the chip: at least 32 k references each)
    const int K = rbytes >= ((size_t)64 << 20) ? (int)std::max<int64_t>(1, std::min<int64_t>(8, n / 32768)) : 1;
    // slices of about equal bytes (the references may be ragged)
    int64_t lo[9]; lo[0] = 0; lo[K] = n;
    for (int sl = 1; sl < K; ++sl) {
        const int64_t target = (int64_t)(rbytes / K) * sl;
        lo[sl] = std::lower_bound(roff, roff + n + 1, target) - roff;
        if (lo[sl] < lo[sl - 1]) lo[sl] = lo[sl - 1];
        if (lo[sl] > n) lo[sl] = n;
    }
    HIP_OR_RET(hipMemcpyAsync(dro.p, roff, sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, s_copy));
    const int wild = profile_has_wildcard(profile);
    if (stats && !(cfg->want & PMX_WANT_SORTED) && rbytes >= ((size_t)64 << 20)) {
        // Statistics of the profile arm are counted along a traceback that already works through the references in chunks of
        // its own (tens of thousands per launch, two launches in flight): cutting the batch into slices as below would shrink
        // those launches.  One call instead; the upload goes in eight slices and every chunk waits only for the slices it reads.
        UploadHook hook; hook.K = 8;
        for (int sl = 0; sl < 8; ++sl) { hook.hi[sl] = n * (sl + 1) / 8; hook.ev[sl] = s_up[sl]; }
        // (copies from pageable memory block the issuing thread: a helper issues them, this thread queues the kernels meanwhile)
        const hipStream_t copy_stream = s_copy;               // (thread-local objects of THIS thread: the helper gets them by value)
        uint8_t *const dr_base = dr.p;
        auto upload = [&hook, copy_stream, dr_base, dev, n, roff, rbuf]() {
            if (hipSetDevice(dev) != hipSuccess) { hook.failed.store(1); return; }
            for (int sl = 0; sl < 8; ++sl) {
                const int64_t a = n * sl / 8, e = n * (sl + 1) / 8;
                hipError_t er = e > a ? hipMemcpyAsync(dr_base + roff[a], rbuf + roff[a], (size_t)(roff[e] - roff[a]), hipMemcpyHostToDevice, copy_stream) : hipSuccess;
                if (er == hipSuccess) er = hipEventRecord(hook.ev[sl], copy_stream);
                if (er != hipSuccess) { hook.failed.store(1); return; }
                hook.recorded.store(sl + 1, std::memory_order_release);
            }
        };
        std::thread up;
        try { up = std::thread(upload); } catch (const std::system_error &) { upload(); }     // no helper: the copies are issued first
        g_upload = &hook;
        const int rc = run_batch_device(cfg, n, dq.p, nullptr, profile->s1Len, dr.p, dro.p, profile->s1Len, mr,
                                        drec.p, dst.p, s_comp, wild);
        g_upload = nullptr;
        if (up.joinable()) up.join();
        if (!rc && hook.failed.load()) { (void)hipStreamSynchronize(s_comp); set_err("upload of the references failed"); return -1; }
        if (rc) { (void)hipStreamSynchronize(s_comp); (void)hipStreamSynchronize(s_copy); return rc; }
        HIP_OR_RET(hipStreamSynchronize(s_comp));
        HIP_OR_RET(hipMemcpy(out, drec.p, sizeof(pmx_record_t) * (size_t)n, hipMemcpyDeviceToHost));
        HIP_OR_RET(hipMemcpy(stats_out, dst.p, sizeof(pmx_stats_t) * (size_t)n, hipMemcpyDeviceToHost));
        return 0;
    }
    for (int sl = 0; sl < K; ++sl) {
        const int64_t a = lo[sl], e = lo[sl + 1];
        if (e <= a) continue;
        HIP_OR_RET(hipMemcpyAsync(dr.p + roff[a], rbuf + roff[a], (size_t)(roff[e] - roff[a]), hipMemcpyHostToDevice, s_copy));
        HIP_OR_RET(hipEventRecord(s_up[sl], s_copy));
        HIP_OR_RET(hipStreamWaitEvent(s_comp, s_up[sl], 0));
        const int rc = run_batch_device(cfg, e - a, dq.p, nullptr, profile->s1Len, dr.p, dro.p + a, profile->s1Len, mr,
                                        drec.p + a, stats ? dst.p + a : nullptr, s_comp, wild);
        if (rc) { (void)hipStreamSynchronize(s_comp); return rc; }
        HIP_OR_RET(hipEventRecord(s_done[sl], s_comp));
    }
    for (int sl = 0; sl < K; ++sl) {
        const int64_t a = lo[sl], e = lo[sl + 1];
        if (e <= a) continue;
        HIP_OR_RET(hipEventSynchronize(s_done[sl]));
        HIP_OR_RET(hipMemcpy(out + a, drec.p + a, sizeof(pmx_record_t) * (size_t)(e - a), hipMemcpyDeviceToHost));
        if (stats) HIP_OR_RET(hipMemcpy(stats_out + a, dst.p + a, sizeof(pmx_stats_t) * (size_t)(e - a), hipMemcpyDeviceToHost));
    }
    return 0;
}

// Device-resident references against one reused query profile, asynchronous on `stream`.
extern "C" int pmx_align_profile_batch_device(const pmx_config_t *cfg, const parasail_profile_t *profile, int64_t n,
                                              const uint8_t *d_rbuf, const int64_t *d_roff, int32_t max_rlen,
                                              pmx_record_t *d_out, pmx_stats_t *d_stats_out, void *stream)
{
    if (check_cfg(cfg)) return -1;
    if (!profile) { set_err("null profile"); return -1; }
    if (profile->matrix != cfg->matrix) { set_err("profile was built with a different matrix"); return -1; }
    if (cfg->matrix->type == PARASAIL_MATRIX_TYPE_PSSM) { set_err("PSSM matrices are single-pair only"); return -1; }
    const uint8_t *dq = nullptr;
    if (profile_device_query(profile, &dq)) return -1;
    StreamGuard guard(stream);
    if (!guard.ok) { set_err("stream guard failed"); return -1; }
    return run_batch_device(cfg, n, dq, nullptr, profile->s1Len, d_rbuf, d_roff, profile->s1Len, max_rlen,
                            d_out, d_stats_out, stream, profile_has_wildcard(profile));
}


// ---- banded batches (extension) ------------------------------------------------------------------------------
// The reference has one banded entry, Aligner::banded_nw -> parasail_nw_banded (src/aligner/mod.rs:454-489: global, main
// diagonal).  The batch form takes any mode and an optional per-pair band centre: cell (i, j) belongs to the band iff
// |(j - i) - diag[pair]| <= band.  BASELINE config 5's "banded SW" is this with mode = local and diag = end_ref - end_query of a
// first full pass (or a seed's diagonal).  Rule and oracle: oracle/pmx_oracle.c:orc_align_ex.
static int banded_device(const pmx_config_t *cfg, int64_t n, const uint8_t *d_qbuf, const int64_t *d_qoff, int q_shared,
                         const uint8_t *d_rbuf, const int64_t *d_roff, int32_t max_qlen, int32_t max_rlen,
                         int32_t band, const int32_t *d_diag, pmx_record_t *d_out, void *stream)
{
    if (check_cfg(cfg)) return -1;
    if (n <= 0) return 0;
    if (band < 0) { set_err("band must be >= 0"); return -1; }
    if (max_qlen <= 0 || max_rlen <= 0) { set_err("max_qlen / max_rlen must be positive"); return -1; }
    if (cfg->want & ~PMX_WANT_SORTED) { set_err("banded batches return score and end positions only"); return -1; }
    if (cfg->matrix->type == PARASAIL_MATRIX_TYPE_PSSM) { set_err("PSSM matrices are single-pair only"); return -1; }
    DevMat dm;
    if (get_devmat(cfg->matrix, &dm)) return -1;
    StreamGuard guard(stream);
    if (!guard.ok) { set_err("stream guard failed"); return -1; }
    pmx_config_t c = *cfg; c.width = 32;                       // 32-bit lanes: no saturation inside a band
    const char *kname = "pmx_banded_kernel";
    void *sort_scr = nullptr, *retry_scr = nullptr;
    if (n >= 4096 && n < (1LL << 32) && scratch_reserve(pmx_sort_scratch_bytes(n), &sort_scr, SCR_SORT)) return -1;
    if (n < (1LL << 31) && scratch_reserve(2 * ((size_t)n + 1) * sizeof(unsigned), &retry_scr, SCR_RETRY)) return -1;      // two lists: wildcards, ties
    const int rcb = pmx_launch_banded(c.mode, c.sg_flags, c.open, c.extend, dm.d, n, d_qbuf, d_qoff, q_shared, d_rbuf, d_roff,
                                      max_qlen, max_rlen, band, d_diag, d_out, (hipStream_t)stream, &kname, sort_scr,
                                      retry_scr ? (unsigned *)retry_scr + 1 : nullptr, (int *)retry_scr);
    if (rcb < 0) { set_err("banded kernel launch failed: %s", hipGetErrorString((hipError_t)(-rcb))); return rcb; }
    if (rcb == 0) { g_last_kernel = kname; return 0; }
    const int rc = general_batch(&c, dm, n, d_qbuf, d_qoff, q_shared, d_rbuf, d_roff, max_rlen, band, d_diag, false, d_out, nullptr,
                                 (hipStream_t)stream, max_qlen);
    if (rc == 0) g_last_kernel = "pmx_general_kernel/banded";
    return rc;
}

extern "C" int pmx_align_batch_banded_device(const pmx_config_t *cfg, int64_t n,
                                             const uint8_t *d_qbuf, const int64_t *d_qoff,
                                             const uint8_t *d_rbuf, const int64_t *d_roff,
                                             int32_t max_qlen, int32_t max_rlen, int32_t band, const int32_t *d_diag,
                                             pmx_record_t *d_out, void *stream)
{
    if (!d_qbuf || !d_qoff || !d_rbuf || !d_roff || !d_out) { set_err("null buffer"); return -1; }
    return banded_device(cfg, n, d_qbuf, d_qoff, 0, d_rbuf, d_roff, max_qlen, max_rlen, band, d_diag, d_out, stream);
}

extern "C" int pmx_align_profile_batch_banded_device(const pmx_config_t *cfg, const parasail_profile_t *profile, int64_t n,
                                                     const uint8_t *d_rbuf, const int64_t *d_roff, int32_t max_rlen,
                                                     int32_t band, const int32_t *d_diag, pmx_record_t *d_out, void *stream)
{
    if (!profile) { set_err("null profile"); return -1; }
    if (!cfg || profile->matrix != cfg->matrix) { set_err("profile was built with a different matrix"); return -1; }
    if (!d_rbuf || !d_roff || !d_out) { set_err("null buffer"); return -1; }
    const uint8_t *dq = nullptr;
    if (profile_device_query(profile, &dq)) return -1;
    return banded_device(cfg, n, dq, nullptr, profile->s1Len, d_rbuf, d_roff, profile->s1Len, max_rlen, band, d_diag, d_out, stream);
}

// Host buffers in, host records out.  profile != NULL: the profile arm (qbuf / qoff are ignored).
extern "C" int pmx_align_batch_banded(const pmx_config_t *cfg, const parasail_profile_t *profile, int64_t n,
                                      const uint8_t *qbuf, const int64_t *qoff, const uint8_t *rbuf, const int64_t *roff,
                                      int32_t band, const int32_t *diag, pmx_record_t *out)
{
    if (check_cfg(cfg)) return -1;
    if (n <= 0) return 0;
    if (!rbuf || !roff || !out || (!profile && (!qbuf || !qoff))) { set_err("null buffer"); return -1; }
    int32_t mq = 0, mr = 0; bool bad = false;
    host_maxlens(n, roff, &mr, &bad);
    if (!profile) host_maxlens(n, qoff, &mq, &bad); else mq = profile->s1Len;
    if (bad || roff[0] != 0 || (!profile && qoff[0] != 0)) { set_err("bad offsets (every sequence needs length >= 1, offsets start at 0)"); return -1; }
    DevBuf<uint8_t> dq, dr; DevBuf<int64_t> dqo, dro; DevBuf<int32_t> dd; DevBuf<pmx_record_t> drec;
    if (dr.try_alloc((size_t)roff[n]) || dro.try_alloc(n + 1) || drec.try_alloc(n) || (diag && dd.try_alloc(n)) ||
        (!profile && (dq.try_alloc((size_t)qoff[n]) || dqo.try_alloc(n + 1)))) { set_err("out of device memory"); return -2; }
    HIP_OR_RET(hipMemcpy(dr.p, rbuf, (size_t)roff[n], hipMemcpyHostToDevice));
    HIP_OR_RET(hipMemcpy(dro.p, roff, sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice));
    if (diag) HIP_OR_RET(hipMemcpy(dd.p, diag, sizeof(int32_t) * n, hipMemcpyHostToDevice));
    int rc;
    if (profile) rc = pmx_align_profile_batch_banded_device(cfg, profile, n, dr.p, dro.p, mr, band, diag ? dd.p : nullptr, drec.p, nullptr);
    else {
        HIP_OR_RET(hipMemcpy(dq.p, qbuf, (size_t)qoff[n], hipMemcpyHostToDevice));
        HIP_OR_RET(hipMemcpy(dqo.p, qoff, sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice));
        rc = pmx_align_batch_banded_device(cfg, n, dq.p, dqo.p, dr.p, dro.p, mq, mr, band, diag ? dd.p : nullptr, drec.p, nullptr);
    }
    if (rc) return rc;
    HIP_OR_RET(hipMemcpy(out, drec.p, sizeof(pmx_record_t) * n, hipMemcpyDeviceToHost));
    return 0;
}

// ---- score tables for a batch (extension; the reference returns one table per call, src/alignment/mod.rs:123-192) ----------
// d_tab_off[k] = number of cells before pair k's [qlen][rlen] int32 table in d_score_table (n + 1 entries); d_score_row is packed
// like the references (roff), d_score_col like the queries (qoff); any of the three outputs may be NULL.
extern "C" int pmx_align_batch_table_device(const pmx_config_t *cfg, int64_t n,
                                            const uint8_t *d_qbuf, const int64_t *d_qoff,
                                            const uint8_t *d_rbuf, const int64_t *d_roff,
                                            int32_t max_qlen, int32_t max_rlen,
                                            const int64_t *d_tab_off, int32_t *d_score_table,
                                            int32_t *d_score_row, int32_t *d_score_col,
                                            pmx_record_t *d_out, void *stream)
{
    if (check_cfg(cfg)) return -1;
    if (n <= 0) return 0;
    if (!d_qbuf || !d_qoff || !d_rbuf || !d_roff) { set_err("null buffer"); return -1; }
    if (d_score_table && !d_tab_off) { set_err("a score table needs d_tab_off"); return -1; }
    if (max_qlen <= 0 || max_rlen <= 0) { set_err("max_qlen / max_rlen must be positive"); return -1; }
    if (cfg->want & ~PMX_WANT_SORTED) { set_err("table batches return score tables, rows / columns and records"); return -1; }
    if (cfg->matrix->type == PARASAIL_MATRIX_TYPE_PSSM) { set_err("PSSM matrices are single-pair only"); return -1; }
    DevMat dm;
    if (get_devmat(cfg->matrix, &dm)) return -1;
    StreamGuard guard(stream);
    if (!guard.ok) { set_err("stream guard failed"); return -1; }
    hipStream_t st = (hipStream_t)stream;
    int rc = pmx_launch_table(cfg->mode, cfg->sg_flags, cfg->open, cfg->extend, dm.d, n, d_qbuf, d_qoff, 0, d_rbuf, d_roff,
                              max_qlen, max_rlen, d_tab_off, d_score_table, d_score_row, d_score_col, d_out, st);
    if (rc < 0) { set_err("table kernel launch failed: %s", hipGetErrorString((hipError_t)(-rc))); return rc; }
    if (rc == 0) { g_last_kernel = "pmx_table_kernel"; return 0; }
    // outside the row-by-row kernel's window (references beyond 1 024 symbols, open < extend, ...): the general kernel, in chunks
    const size_t stride = (size_t)8 * max_rlen;
    const bool fits = pmx_general_lds_fits(dm.d.msize, dm.d.msize, max_rlen);
    const size_t rs_stride = fits ? 0 : (((size_t)max_rlen + 8 + 15) & ~(size_t)15);
    const size_t per_pair = stride * sizeof(int32_t) + rs_stride;
    const char *cb = pmx_env("PMX_GENERAL_CHUNK_BYTES");
    int64_t chunk = (int64_t)((cb && atof(cb) > 0 ? atof(cb) : 2e9) / (double)per_pair);
    if (chunk < 1) chunk = 1;
    if (chunk > n) chunk = n;
    void *bound = nullptr;
    if (scratch_reserve((size_t)chunk * per_pair, &bound)) return -1;
    DevBuf<pmx_record_t> tmp_rec;
    if (!d_out && tmp_rec.try_alloc((size_t)n)) { set_err("out of device memory"); return -2; }
    for (int64_t c0 = 0; c0 < n; c0 += chunk) {
        const int64_t m = (n - c0 < chunk) ? n - c0 : chunk;
        PmxGeneralArgs a; memset(&a, 0, sizeof a);
        a.qbuf = d_qbuf; a.qoff = d_qoff + c0; a.rbuf = d_rbuf; a.roff = d_roff + c0; a.n = m; a.max_rlen = max_rlen;
        a.scores = dm.d.scores; a.mapper = dm.d.mapper; a.msize = dm.d.msize; a.mat_rows = dm.d.msize;
        a.mode = cfg->mode; a.sg_flags = cfg->sg_flags; a.open = cfg->open; a.ext = cfg->extend; a.band = -1; a.bits = 32;
        a.bound = (int32_t *)bound; a.bound_stride = (long long)stride;
        if (!fits) { a.rs_scratch = (uint8_t *)bound + (size_t)chunk * stride * sizeof(int32_t); a.rs_stride = (long long)rs_stride; }
        a.rec = (d_out ? d_out : tmp_rec.p) + c0;
        a.tab_off = d_tab_off ? d_tab_off + c0 : nullptr; a.score_table = d_score_table;
        a.score_row = d_score_row; a.score_col = d_score_col;
        rc = pmx_launch_general(a, false, st);
        if (rc) { set_err("general kernel launch failed (%d)", rc); return rc < 0 ? rc : -1; }
    }
    if (!d_out) HIP_OR_RET(hipStreamSynchronize(st));
    g_last_kernel = "pmx_general_kernel/tables";
    return 0;
}

// ---- device-resident CIGAR entry ------------------------------------------------------------------------
// Sweep (packed 4-bit traceback to HBM scratch) and walk run in chunks on two streams: the walk of chunk c (latency-bound, one
// lane per pair) runs beside the sweep of chunk c + 1 (VALU-bound); the trace scratch is double-buffered.  The walk leaves
// run-length ops in per-pair slots and each pair's text length; one scan and one render finish the batch on the caller's stream.
// 0 done (asynchronously on `st`), 1 not eligible for the packed traceback sweeps, <0 error.
// The offset arrays are absolute into d_qbuf / d_rbuf; ops_base = qoff[0] + roff[0] (0 when the offsets start at 0).
static int cigar_device_run(const pmx_config_t *cfg, const DevMat &dm, int64_t n,
                            const uint8_t *d_qbuf, const int64_t *d_qoff, const uint8_t *d_rbuf, const int64_t *d_roff,
                            int32_t mq, int32_t mr, long long ops_base,
                            pmx_record_t *d_out, char *d_text, int64_t capacity, int64_t *d_text_off, hipStream_t st)
{
    if (cfg->width == 8 || cfg->matrix->type != PARASAIL_MATRIX_TYPE_SQUARE) return 1;
    PmxBatch b = {d_qbuf, d_qoff, d_rbuf, d_roff, n, mq, mr, 0, nullptr, nullptr, nullptr, 0, 0};
    int variant = 0, Tmax = 0; size_t tbytes = 0;
    if (pmx_trace16_plan(b, dm.d, cfg->mode, cfg->open, cfg->extend, &variant, &Tmax, &tbytes) != 0 || variant < 10) return 1;
    if (trace_ws_init()) return -1;
    // chunks: at most ~12 GB of trace each (two buffers; measured on 1.25 M pairs of 250 x 250: 3 GB chunks 27.8 ms, 12 GB 26.2 ms --
    // fewer launch tails), at most 15 % of the free HBM each, at least two for the overlap once the batch is worth it
    double chunk_bytes = 12e9;
    { size_t fb = 0, tb = 0; if (hipMemGetInfo(&fb, &tb) == hipSuccess && 0.15 * (double)fb < chunk_bytes) chunk_bytes = 0.15 * (double)fb; }
    if (const char *e = pmx_env("PMX_CIGAR_CHUNK_BYTES")) chunk_bytes = atof(e);      // tests force small chunks
    int64_t nchunks = (int64_t)((double)tbytes / chunk_bytes) + 1;
    if (nchunks < 2 && n >= 16384) nchunks = 2;
    int64_t chunk = ((n + nchunks - 1) / nchunks + 63) / 64 * 64;
    if (chunk > n) chunk = n;
    PmxBatch bc = b; bc.n = chunk;
    size_t cbytes = 0;
    (void)pmx_trace16_plan(bc, dm.d, cfg->mode, cfg->open, cfg->extend, &variant, &Tmax, &cbytes);
    cbytes = (cbytes + 255) & ~(size_t)255;
    const bool two = chunk < n && !pmx_env("PMX_CIGAR_NO_OVERLAP");     // (diagnostics: sweep and walk back to back on one stream)
    uint32_t *tbuf = nullptr, *dops = nullptr; unsigned char *misc = nullptr;
    const size_t scan_bytes = pmx_text_scan_scratch_bytes(n);
    const size_t fstride = (size_t)chunk / 2 + 16;           // per-block flags of a chunk's sweep (perm-table form / LDS-profile form), two sets
    const size_t misc_bytes = (size_t)(4 * n + 2) * sizeof(int32_t) + 256 + scan_bytes + 256 + 2 * fstride * sizeof(int);
    if (scratch_reserve(cbytes * (two ? 2 : 1), (void **)&tbuf, SCR_TRACE) ||
        scratch_reserve((size_t)n * ((size_t)mq + mr + 1) * sizeof(uint32_t), (void **)&dops, SCR_OPS) ||
        scratch_reserve(misc_bytes, (void **)&misc, SCR_CIG)) return -1;
    int32_t *nops = (int32_t *)misc, *beg = nops + n, *textlen = beg + 2 * n;
    void *scan_tmp = (void *)(((uintptr_t)(textlen + n + 2) + 255) & ~(uintptr_t)255);
    int *bflags = (int *)(((uintptr_t)scan_tmp + scan_bytes + 255) & ~(uintptr_t)255);
    // sweeps of consecutive chunks alternate between the caller's stream and an internal one (the tail of one launch is
    // backfilled by the next); every walk runs on the high-priority walk stream after its sweep
    if (two) { HIP_OR_RET(hipEventRecord(g_tws.start, st)); HIP_OR_RET(hipStreamWaitEvent(g_tws.aux, g_tws.start, 0)); }
    int idx = 0;
    for (int64_t c0 = 0; c0 < n; c0 += chunk, ++idx) {
        PmxBatch bk = b;
        bk.n = (n - c0 < chunk) ? n - c0 : chunk;
        bk.qoff = d_qoff + c0; bk.roff = d_roff + c0;
        bk.blockflag = bflags + (size_t)(idx & 1) * fstride;
        const hipStream_t sws = (two && (idx & 1)) ? g_tws.aux : st;
        if (two && idx >= 2) HIP_OR_RET(hipStreamWaitEvent(sws, g_tws.walk_done[idx & 1], 0));     // this trace buffer's last walk is done
        PmxWalkSplit sp = {two ? g_tws.walk : st, g_tws.sweep_done[idx & 1], two ? g_tws.walk_done[idx & 1] : nullptr,
                           ops_base - c0, textlen + c0};
        const int rc = pmx_launch_trace16(variant, bk, dm.d, cfg->mode, cfg->sg_flags, cfg->open, cfg->extend, d_out + c0,
                                          (uint32_t *)((unsigned char *)tbuf + (two ? (size_t)(idx & 1) * cbytes : 0)), Tmax,
                                          dops, nullptr, nops + c0, beg + 2 * c0, sws, nullptr, &sp);
        if (rc) { set_err("traceback launch failed (%d)", rc); return rc < 0 ? rc : -1; }
    }
    if (two) HIP_OR_RET(hipStreamWaitEvent(st, g_tws.walk_done[(idx - 1) & 1], 0));      // the walk stream is in order: the last walk covers all
    int rc = pmx_launch_text_offsets(textlen, n, d_text_off, scan_tmp, scan_bytes, st);
    if (rc) { set_err("text offset scan failed (%d)", rc); return rc; }
    rc = pmx_launch_cigar_render_slots(dops, d_qoff, d_roff, ops_base, nops, d_text_off, d_text, capacity, n, st);
    if (rc) { set_err("cigar render launch failed (%d)", rc); return rc; }
    g_last_kernel = variant >= 20 ? "pmx_sw16_kernel/packed trace + pmx_walkp_kernel" : "pmx_nwsg16v_kernel/packed trace + pmx_walkp_kernel";
    return 0;
}

extern "C" int pmx_align_batch_cigar_device(const pmx_config_t *cfg, int64_t n,
                                            const uint8_t *d_qbuf, const int64_t *d_qoff,
                                            const uint8_t *d_rbuf, const int64_t *d_roff,
                                            int32_t max_qlen, int32_t max_rlen,
                                            pmx_record_t *d_out, char *d_cigar_text, int64_t cigar_capacity,
                                            int64_t *d_cigar_off, void *stream)
{
    if (check_cfg(cfg)) return -1;
    if (n <= 0) return 0;
    if (!d_qbuf || !d_qoff || !d_rbuf || !d_roff || !d_out || !d_cigar_text || !d_cigar_off) { set_err("null buffer"); return -1; }
    if (max_qlen <= 0 || max_rlen <= 0) { set_err("max_qlen / max_rlen must be positive"); return -1; }
    DevMat dm;
    if (get_devmat(cfg->matrix, &dm)) return -1;
    StreamGuard guard(stream);
    if (!guard.ok) { set_err("stream guard failed"); return -1; }
    const int rc = cigar_device_run(cfg, dm, n, d_qbuf, d_qoff, d_rbuf, d_roff, max_qlen, max_rlen, 0, d_out,
                                    d_cigar_text, cigar_capacity, d_cigar_off, (hipStream_t)stream);
    if (rc == 1) set_err("this configuration has no device-resident CIGAR path (width 8, PSSM, open < extend, a matrix whose score + open "
                         "leaves a byte, or queries beyond 1023 symbols): use pmx_align_batch_cigar");
    return rc == 1 ? -1 : rc;
}

// The caller-owned CIGAR text: a malloc block that grows chunk by chunk; device text is copied straight into it.
// Large text blocks go back to a small pool when the caller releases them with pmx_free(), and the next batch call starts from
// one: a caller that aligns batch after batch writes into memory that is already paged in (first-touch faults of a fresh
// 100 MB block cost milliseconds), and the block usually has the right size at once.  At most two blocks, at most 1 GB.
struct TextPool {
    std::mutex mx;
    std::unordered_map<void *, size_t> live;      // blocks handed to callers (capacity)
    std::vector<std::pair<char *, size_t>> idle;  // blocks given back
    static constexpr size_t MIN_BLOCK = 1 << 20, MAX_IDLE_BYTES = (size_t)1 << 30;
    char *take(size_t *cap)
    {
        std::lock_guard<std::mutex> lk(mx);
        if (idle.empty()) return nullptr;
        size_t best = 0;
        for (size_t k = 1; k < idle.size(); ++k) if (idle[k].second > idle[best].second) best = k;
        char *p = idle[best].first; *cap = idle[best].second;
        idle.erase(idle.begin() + (long)best);
        return p;
    }
    void handed_out(void *p, size_t cap) { if (cap >= MIN_BLOCK) { std::lock_guard<std::mutex> lk(mx); live[p] = cap; } }
    bool give_back(void *p)                        // true: the pool keeps it
    {
        std::lock_guard<std::mutex> lk(mx);
        auto it = live.find(p);
        if (it == live.end()) return false;
        const size_t cap = it->second;
        live.erase(it);
        size_t held = 0;
        for (auto &b : idle) held += b.second;
        if (idle.size() >= 2 || held + cap > MAX_IDLE_BYTES) return false;
        idle.emplace_back((char *)p, cap);
        return true;
    }
};
static TextPool g_text_pool;

// The caller-owned CIGAR text: a malloc block that grows chunk by chunk; device text is copied straight into it.
struct TextBuf {
    char *p = nullptr; size_t len = 0, cap = 0;
    char *grow(size_t extra)       // room for `extra` more bytes (+ terminator); returns the write position or nullptr
    {
        if (!p && extra + 1 >= TextPool::MIN_BLOCK / 2) p = g_text_pool.take(&cap);
        if (len + extra + 1 > cap) {
            size_t ncap = cap ? cap * 2 : 4096;
            while (ncap < len + extra + 1) ncap *= 2;
            char *np = (char *)realloc(p, ncap);
            if (!np) return nullptr;
            p = np; cap = ncap;
        }
        return p + len;
    }
};

// CIGAR for a batch.  Fast path: pmx_trace16 (4-bit trace in HBM, on-device walk); otherwise the general
// kernel with byte trace tables and pmx_walk_kernel.  Only the run-length ops come back to the host, which
// renders the text.  One chunk = one set of launches; chunks bound the trace scratch.
// PMX_TIMING=1: stage times of the batch CIGAR entry on stderr
struct StageTimer {
    bool on; double t0; const char *what;
    static double now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
    StageTimer() : on(pmx_env("PMX_TIMING") != nullptr), t0(now()), what("") {}
    void done(const char *stage) { if (on) { (void)hipDeviceSynchronize(); const double t = now(); fprintf(stderr, "[pmx timing] %-28s %8.3f ms\n", stage, (t - t0) * 1e3); t0 = t; } }
};

static int cigar_chunk(const pmx_config_t *cfg, const DevMat &dm, int64_t n,
                       const uint8_t *qbuf, const int64_t *qoff, const uint8_t *rbuf, const int64_t *roff,
                       pmx_record_t *out, TextBuf &text, int64_t *cigar_off /* n+1, cigar_off[0] preset */)
{
    StageTimer tm;
    int32_t mq = 0, mr = 0; bool bad = false;
    host_maxlens(n, qoff, &mq, &bad); host_maxlens(n, roff, &mr, &bad);
    if (bad || qoff[0] != 0 || roff[0] != 0) { set_err("bad offsets"); return -1; }
    std::vector<int64_t> ops_off(n + 1);
    ops_off[0] = 0;
    for (int64_t k = 0; k < n; ++k) ops_off[k + 1] = ops_off[k] + (qoff[k + 1] - qoff[k]) + (roff[k + 1] - roff[k]) + 1;
    const size_t qbytes = (size_t)qoff[n], rbytes = (size_t)roff[n];
    DevBuf<uint8_t> dq, dr; DevBuf<int64_t> dqo, dro, doo; DevBuf<pmx_record_t> drec; DevBuf<int32_t> dnops, dbeg;
    if (dq.try_alloc(qbytes) || dr.try_alloc(rbytes) || dqo.try_alloc(n + 1) || dro.try_alloc(n + 1) || doo.try_alloc(n + 1) ||
        drec.try_alloc(n) || dnops.try_alloc(n) || dbeg.try_alloc(2 * n)) { set_err("out of device memory"); return -2; }
    uint32_t *dops = nullptr;
    if (scratch_reserve((size_t)ops_off[n] * sizeof(uint32_t), (void **)&dops, SCR_OPS)) return -1;
    tm.done("host prep + device alloc");
    HIP_OR_RET(hipMemcpy(dq.p, qbuf, qbytes, hipMemcpyHostToDevice));
    HIP_OR_RET(hipMemcpy(dr.p, rbuf, rbytes, hipMemcpyHostToDevice));
    HIP_OR_RET(hipMemcpy(dqo.p, qoff, sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice));
    HIP_OR_RET(hipMemcpy(dro.p, roff, sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice));
    HIP_OR_RET(hipMemcpy(doo.p, ops_off.data(), sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice));

    tm.done("H2D");
    PmxBatch b = {dq.p, dqo.p, dr.p, dro.p, n, mq, mr, 0, nullptr, nullptr, nullptr, 0, 0};
    int variant = 0, Tmax = 0; size_t tbytes = 0;
    int rc;
    if (cfg->width != 8 && cfg->matrix->type == PARASAIL_MATRIX_TYPE_SQUARE &&
        pmx_trace16_plan(b, dm.d, cfg->mode, cfg->open, cfg->extend, &variant, &Tmax, &tbytes, false) == 0) {   // (packed sweeps: the pipelined path)
        uint32_t *tbuf = nullptr;
        if (scratch_reserve(tbytes, (void **)&tbuf, SCR_TRACE)) return -1;
        rc = pmx_launch_trace16(variant, b, dm.d, cfg->mode, cfg->sg_flags, cfg->open, cfg->extend, drec.p, tbuf, Tmax,
                                dops, doo.p, dnops.p, dbeg.p, nullptr);
        g_last_kernel = variant >= 20 ? "pmx_sw16_kernel/packed trace + pmx_walkp_kernel" : variant >= 10 ? "pmx_nwsg16v_kernel/packed trace + pmx_walkp_kernel" : "pmx_trace16_kernel + pmx_walk16_kernel";
        if (rc) { set_err("trace16 launch failed (%d)", rc); return rc < 0 ? rc : -1; }
    } else {
        std::vector<int64_t> tab_off(n + 1);
        tab_off[0] = 0;
        for (int64_t k = 0; k < n; ++k) tab_off[k + 1] = tab_off[k] + (qoff[k + 1] - qoff[k]) * (roff[k + 1] - roff[k]);
        DevBuf<int64_t> dto;
        if (dto.try_alloc(n + 1)) { set_err("out of device memory"); return -2; }
        HIP_OR_RET(hipMemcpy(dto.p, tab_off.data(), sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice));
        int8_t *dtrace = nullptr; void *bound = nullptr;
        if (scratch_reserve((size_t)tab_off[n], (void **)&dtrace, SCR_TRACE)) return -1;
        const size_t stride = (size_t)8 * mr;
        const bool fits = pmx_general_lds_fits(dm.d.msize, dm.d.msize, mr);
        const size_t rs_stride = fits ? 0 : (((size_t)mr + 8 + 15) & ~(size_t)15);
        if (scratch_reserve((size_t)n * (stride * sizeof(int32_t) + rs_stride), &bound)) return -1;
        PmxGeneralArgs a; memset(&a, 0, sizeof a);
        if (!fits) { a.rs_scratch = (uint8_t *)bound + (size_t)n * stride * sizeof(int32_t); a.rs_stride = (long long)rs_stride; }
        a.qbuf = dq.p; a.qoff = dqo.p; a.rbuf = dr.p; a.roff = dro.p; a.n = n; a.max_rlen = mr;
        a.scores = dm.d.scores; a.mapper = dm.d.mapper; a.msize = dm.d.msize; a.mat_rows = dm.d.msize;
        a.mode = cfg->mode; a.sg_flags = cfg->sg_flags; a.open = cfg->open; a.ext = cfg->extend; a.band = -1;
        a.bits = cfg->width;
        a.bound = (int32_t *)bound; a.bound_stride = (long long)stride;
        a.rec = drec.p; a.tab_off = dto.p; a.trace_table = dtrace;
        rc = pmx_launch_general(a, false, nullptr);
        if (rc) { set_err("general kernel launch failed (%d)", rc); return rc < 0 ? rc : -1; }
        PmxWalkArgs w; memset(&w, 0, sizeof w);
        w.qbuf = dq.p; w.qoff = dqo.p; w.rbuf = dr.p; w.roff = dro.p; w.n = n;
        w.mapper = dm.d.mapper; w.mode = cfg->mode; w.trace_table = dtrace; w.tab_off = dto.p; w.rec = drec.p;
        w.ops = dops; w.ops_off = doo.p; w.nops = dnops.p; w.beg = dbeg.p;
        rc = pmx_launch_walk(w, nullptr);
        g_last_kernel = "pmx_general_kernel + pmx_walk_kernel";
        if (rc) { set_err("walk kernel launch failed (%d)", rc); return rc < 0 ? rc : -1; }
        HIP_OR_RET(hipDeviceSynchronize());      // dto is released on scope exit
    }
    tm.done("sweep + walk kernels");
    // The CIGAR text is rendered on the device: text lengths come back (4 bytes per pair), the host turns them
    // into offsets, the text itself is written there and copied back in one piece.
    HIP_OR_RET(hipMemcpy(out, drec.p, sizeof(pmx_record_t) * n, hipMemcpyDeviceToHost));
    DevBuf<int32_t> dtl;
    if (dtl.try_alloc(n)) { set_err("out of device memory"); return -2; }
    rc = pmx_launch_cigar_textlen(dops, doo.p, dnops.p, dtl.p, n, nullptr);
    if (rc) { set_err("cigar length kernel launch failed (%d)", rc); return rc < 0 ? rc : -1; }
    std::vector<int32_t> tl(n);
    HIP_OR_RET(hipMemcpy(tl.data(), dtl.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    std::vector<int64_t> toff(n + 1);
    toff[0] = 0;
    for (int64_t k = 0; k < n; ++k) toff[k + 1] = toff[k] + tl[k];
    DevBuf<int64_t> dtoff; DevBuf<char> dtext;
    if (dtoff.try_alloc(n + 1) || dtext.try_alloc((size_t)toff[n] + 1)) { set_err("out of device memory"); return -2; }
    HIP_OR_RET(hipMemcpy(dtoff.p, toff.data(), sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice));
    rc = pmx_launch_cigar_render(dops, doo.p, dnops.p, dtoff.p, dtext.p, n, nullptr);
    if (rc) { set_err("cigar render kernel launch failed (%d)", rc); return rc < 0 ? rc : -1; }
    const size_t base = text.len;
    char *dst = text.grow((size_t)toff[n]);
    if (!dst) { set_err("out of memory"); return -1; }
    if (toff[n]) HIP_OR_RET(hipMemcpy(dst, dtext.p, (size_t)toff[n], hipMemcpyDeviceToHost));
    text.len += (size_t)toff[n];
    for (int64_t k = 0; k < n; ++k) cigar_off[k + 1] = (int64_t)base + toff[k + 1];
    tm.done("render + D2H");
    return 0;
}

// Host entry on top of the device entry: the sequence bytes go up in slices on a copy stream, every slice runs the device
// pipeline (sweep / walk overlapped inside) on a compute stream as soon as its bytes have arrived, and the host copies a finished
// slice's records, offsets and text back while the following slices compute.  0 done, 1 not eligible, <0 error.
static int cigar_host_pipelined(const pmx_config_t *cfg, const DevMat &dm, int64_t n,
                                const uint8_t *qbuf, const int64_t *qoff, const uint8_t *rbuf, const int64_t *roff,
                                pmx_record_t *out, TextBuf &text, int64_t *cigar_off)
{
    StageTimer tm;
    int32_t mq = 0, mr = 0; bool bad = false;
    host_maxlens(n, qoff, &mq, &bad); host_maxlens(n, roff, &mr, &bad);
    if (bad) { set_err("every sequence must have length >= 1"); return -1; }
    {   // eligibility before anything is staged
        PmxBatch b = {nullptr, nullptr, nullptr, nullptr, n, mq, mr, 0, nullptr, nullptr, nullptr, 0, 0};
        int variant = 0, Tmax = 0; size_t tbytes = 0;
        if (cfg->width == 8 || pmx_trace16_plan(b, dm.d, cfg->mode, cfg->open, cfg->extend, &variant, &Tmax, &tbytes) != 0 || variant < 10) return 1;
    }
    static thread_local hipStream_t s_copy = nullptr, s_comp = nullptr;
    static thread_local hipEvent_t s_up[8], s_done[8];
    static thread_local int s_dev = -1;
    int dev = 0; HIP_OR_RET(hipGetDevice(&dev));
    if (s_dev != dev) {
        if (s_copy) { (void)hipStreamDestroy(s_copy); (void)hipStreamDestroy(s_comp); for (int k = 0; k < 8; ++k) { (void)hipEventDestroy(s_up[k]); (void)hipEventDestroy(s_done[k]); } }
        HIP_OR_RET(hipStreamCreateWithFlags(&s_copy, hipStreamNonBlocking));
        HIP_OR_RET(hipStreamCreateWithFlags(&s_comp, hipStreamNonBlocking));
        for (int k = 0; k < 8; ++k) {
            HIP_OR_RET(hipEventCreateWithFlags(&s_up[k], hipEventDisableTiming));
            HIP_OR_RET(hipEventCreateWithFlags(&s_done[k], hipEventDisableTiming));
        }
        s_dev = dev;
    }
    const int K = n >= 262144 ? 8 : n >= 32768 ? 2 : 1;
    const size_t qbytes = (size_t)qoff[n], rbytes = (size_t)roff[n];
    // text capacity per slice: half a byte per sequence symbol + 16 per pair covers related reads many times over; a slice
    // that needs more is rendered again into an exact-size buffer (the ops are still in the scratch)
    int64_t cap[8], tbase[8], lo[8], hi[8];
    int64_t cap_total = 0;
    for (int sl = 0; sl < K; ++sl) {
        lo[sl] = n * sl / K; hi[sl] = n * (sl + 1) / K;
        cap[sl] = ((qoff[hi[sl]] - qoff[lo[sl]]) + (roff[hi[sl]] - roff[lo[sl]])) / 2 + 16 * (hi[sl] - lo[sl]) + 256;
        cap[sl] = (cap[sl] + 255) & ~(int64_t)255;
        tbase[sl] = cap_total; cap_total += cap[sl];
    }
    uint8_t *dq, *dr; int64_t *dqo, *dro, *dtoff; pmx_record_t *drec; char *dtext;
    if (scratch_reserve(qbytes, (void **)&dq, SCR_HQ) || scratch_reserve(rbytes, (void **)&dr, SCR_HR) ||
        scratch_reserve(sizeof(int64_t) * (n + 1), (void **)&dqo, SCR_HQO) || scratch_reserve(sizeof(int64_t) * (n + 1), (void **)&dro, SCR_HRO) ||
        scratch_reserve(sizeof(pmx_record_t) * n, (void **)&drec, SCR_HREC) ||
        scratch_reserve((size_t)cap_total, (void **)&dtext, SCR_HTEXT) ||
        scratch_reserve(sizeof(int64_t) * (n + K), (void **)&dtoff, SCR_HTOFF)) return -1;
    HIP_OR_RET(hipMemcpyAsync(dqo, qoff, sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, s_copy));
    HIP_OR_RET(hipMemcpyAsync(dro, roff, sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, s_copy));
    for (int sl = 0; sl < K; ++sl) {
        const int64_t a = lo[sl], e = hi[sl];
        if (e <= a) continue;
        HIP_OR_RET(hipMemcpyAsync(dq + qoff[a], qbuf + qoff[a], (size_t)(qoff[e] - qoff[a]), hipMemcpyHostToDevice, s_copy));
        HIP_OR_RET(hipMemcpyAsync(dr + roff[a], rbuf + roff[a], (size_t)(roff[e] - roff[a]), hipMemcpyHostToDevice, s_copy));
        HIP_OR_RET(hipEventRecord(s_up[sl], s_copy));
        HIP_OR_RET(hipStreamWaitEvent(s_comp, s_up[sl], 0));
        const int rc = cigar_device_run(cfg, dm, e - a, dq, dqo + a, dr, dro + a, mq, mr, (long long)(qoff[a] + roff[a]),
                                        drec + a, dtext + tbase[sl], cap[sl], dtoff + a + sl, s_comp);
        if (rc) { (void)hipStreamSynchronize(s_comp); return rc; }
        HIP_OR_RET(hipEventRecord(s_done[sl], s_comp));
    }
    tm.done("queue H2D + kernels");
    std::vector<int64_t> toff;
    for (int sl = 0; sl < K; ++sl) {
        const int64_t a = lo[sl], e = hi[sl], m = e - a;
        if (m <= 0) continue;
        HIP_OR_RET(hipEventSynchronize(s_done[sl]));
        toff.resize((size_t)m + 1);
        HIP_OR_RET(hipMemcpy(toff.data(), dtoff + a + sl, sizeof(int64_t) * (m + 1), hipMemcpyDeviceToHost));
        HIP_OR_RET(hipMemcpy(out + a, drec + a, sizeof(pmx_record_t) * m, hipMemcpyDeviceToHost));
        const int64_t total = toff[m];
        char *dst = text.grow((size_t)total);
        if (!dst) { (void)hipStreamSynchronize(s_comp); set_err("out of memory"); return -1; }
        if (total > cap[sl]) {
            // rare: the slice's text did not fit its share; every later slice has to finish first (the ops scratch is reused per slice),
            // so redo this slice alone with an exact-size text buffer
            HIP_OR_RET(hipStreamSynchronize(s_comp));
            DevBuf<char> big; DevBuf<int64_t> boff;
            if (big.try_alloc((size_t)total + 1) || boff.try_alloc((size_t)m + 1)) { set_err("out of device memory"); return -1; }
            const int rc = cigar_device_run(cfg, dm, m, dq, dqo + a, dr, dro + a, mq, mr, (long long)(qoff[a] + roff[a]),
                                            drec + a, big.p, total, boff.p, s_comp);
            if (rc) return rc < 0 ? rc : -1;
            HIP_OR_RET(hipStreamSynchronize(s_comp));
            HIP_OR_RET(hipMemcpy(dst, big.p, (size_t)total, hipMemcpyDeviceToHost));
        } else if (total) {
            HIP_OR_RET(hipMemcpy(dst, dtext + tbase[sl], (size_t)total, hipMemcpyDeviceToHost));
        }
        const int64_t base = (int64_t)text.len;
        for (int64_t k = 0; k < m; ++k) cigar_off[a + k + 1] = base + toff[k + 1];
        text.len += (size_t)total;
    }
    tm.done("kernels + D2H");
    return 0;
}

extern "C" int pmx_align_batch_cigar(const pmx_config_t *cfg, int64_t n,
                                     const uint8_t *qbuf, const int64_t *qoff,
                                     const uint8_t *rbuf, const int64_t *roff,
                                     pmx_record_t *out, char **cigar_buf, int64_t *cigar_off)
{
    if (check_cfg(cfg)) return -1;
    if (!cigar_buf || !cigar_off) { set_err("null cigar output"); return -1; }
    *cigar_buf = nullptr;
    if (n <= 0) return 0;
    if (cfg->matrix->type == PARASAIL_MATRIX_TYPE_PSSM) { set_err("PSSM matrices are single-pair only"); return -1; }
    if (qoff[0] != 0 || roff[0] != 0) { set_err("offset arrays must start at 0"); return -1; }
    DevMat dm;
    if (get_devmat(cfg->matrix, &dm)) return -1;
    TextBuf text;
    cigar_off[0] = 0;
    {
        const int rc = cigar_host_pipelined(cfg, dm, n, qbuf, qoff, rbuf, roff, out, text, cigar_off);
        if (rc < 0) { free(text.p); return rc; }
        if (rc == 0) {
            if (!text.grow(0)) { set_err("out of memory"); return -1; }
            text.p[text.len] = 0;
            *cigar_buf = text.p; g_text_pool.handed_out(text.p, text.cap);
            return 0;
        }
    }
    // Chunks bound the per-launch trace scratch: budgeted at one byte per cell of the padded tables (the
    // general kernel's layout; the fast kernels write 4 bits per cell).  Large chunks matter: the walk is one
    // lane per pair and hides its dependent-load latency only with many waves in flight.  Up to 96 GB,
    // at most 45 % of the free HBM.
    double chunk_bytes = 96e9;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && 0.45 * (double)free_b < chunk_bytes) chunk_bytes = 0.45 * (double)free_b;
    }
    if (const char *e = pmx_env("PMX_CIGAR_CHUNK_BYTES")) chunk_bytes = atof(e);      // tests force small chunks
    // equal shares: as many chunks as the budget needs, each with about the same number of table bytes
    double total_bytes = 0;
    for (int64_t k = 0; k < n; ++k) total_bytes += 1.0 * (double)(qoff[k + 1] - qoff[k] + 64) * (double)(roff[k + 1] - roff[k] + 64);
    const double nchunks = total_bytes > chunk_bytes ? (double)(int64_t)(total_bytes / chunk_bytes + 1.0) : 1.0;
    const double share = total_bytes / nchunks + 1.0;
    int64_t c0 = 0;
    while (c0 < n) {
        int64_t c1 = c0; double bytes = 0;
        while (c1 < n && (c1 == c0 || bytes < share)) {
            bytes += 1.0 * (double)(qoff[c1 + 1] - qoff[c1] + 64) * (double)(roff[c1 + 1] - roff[c1] + 64);
            ++c1;
        }
        const int64_t m = c1 - c0;
        std::vector<int64_t> qo(m + 1), ro(m + 1);
        for (int64_t k = 0; k <= m; ++k) { qo[k] = qoff[c0 + k] - qoff[c0]; ro[k] = roff[c0 + k] - roff[c0]; }
        const int rc = cigar_chunk(cfg, dm, m, qbuf + qoff[c0], qo.data(), rbuf + roff[c0], ro.data(),
                                   out + c0, text, cigar_off + c0);
        if (rc) { free(text.p); return rc; }
        c0 = c1;
    }
    if (!text.grow(0)) { set_err("out of memory"); return -1; }
    text.p[text.len] = 0;
    *cigar_buf = text.p; g_text_pool.handed_out(text.p, text.cap);
    return 0;
}


// ============================================================================= multi-GPU ===
// Pairs are independent (the reference's only parallel story is user threads sharing a read-only profile,
// tests/test_parasail.rs:689-723), so a batch shards across the GPUs of a node with no data-path collective: a contiguous block
// of pairs per device, cut so that every device gets about the same number of cells (sum of qlen * rlen), one host thread and one
// set of streams per device, results written straight into the caller's arrays in input order -- with one process driving all
// GPUs the per-device D2H copy IS the gather (one process per GPU + an RCCL gather: parasail-rs_amd/sharding.py, bench.py).
extern "C" int pmx_shard_bounds_by_cells(int64_t n, const int64_t *qoff /* NULL: one shared query */, const int64_t *roff,
                                         int parts, int64_t *bounds /* parts + 1 */)
{
    if (n < 0 || parts <= 0 || !roff || !bounds) return -1;
    // cumulative cells (a shared query weighs every reference by the same factor: the reference lengths alone decide)
    std::vector<double> cum((size_t)n + 1);
    cum[0] = 0;
    for (int64_t k = 0; k < n; ++k) {
        const double ql = qoff ? (double)(qoff[k + 1] - qoff[k]) : 1.0;
        cum[k + 1] = cum[k] + ql * (double)(roff[k + 1] - roff[k]);
    }
    bounds[0] = 0;
    for (int g = 1; g < parts; ++g) {
        const double target = cum[n] * (double)g / (double)parts;
        int64_t k = (int64_t)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
        if (k < bounds[g - 1]) k = bounds[g - 1];
        if (k > n) k = n;
        bounds[g] = k;
    }
    bounds[parts] = n;
    return 0;
}

namespace {
struct ShardJob {
    const pmx_config_t *cfg; const parasail_profile_t *profile;
    int64_t lo, hi;
    const uint8_t *qbuf; const int64_t *qoff; const uint8_t *rbuf; const int64_t *roff;
    pmx_record_t *out; pmx_stats_t *stats;
    int device, rc; char err[256];
};
// One persistent host thread per shard slot: its thread-local device scratch, streams and staging survive between calls.
struct ShardWorker {
    std::thread th; std::mutex mx; std::condition_variable cv;
    ShardJob *job = nullptr; bool done = true;
    void loop()
    {
        for (;;) {
            ShardJob *j;
            { std::unique_lock<std::mutex> lk(mx); cv.wait(lk, [&] { return job != nullptr; }); j = job; }
            run(*j);
            { std::lock_guard<std::mutex> lk(mx); job = nullptr; done = true; }
            cv.notify_all();
        }
    }
    static void run(ShardJob &j)
    {
        j.rc = 0; j.err[0] = 0;
        if (hipSetDevice(j.device) != hipSuccess) { j.rc = -1; snprintf(j.err, sizeof j.err, "hipSetDevice(%d) failed", j.device); return; }
        const int64_t m = j.hi - j.lo;
        if (m <= 0) return;
        std::vector<int64_t> ro((size_t)m + 1), qo;
        for (int64_t k = 0; k <= m; ++k) ro[k] = j.roff[j.lo + k] - j.roff[j.lo];
        if (j.profile) {
            j.rc = pmx_align_profile_batch(j.cfg, j.profile, m, j.rbuf + j.roff[j.lo], ro.data(), j.out + j.lo, j.stats ? j.stats + j.lo : nullptr);
        } else {
            qo.resize((size_t)m + 1);
            for (int64_t k = 0; k <= m; ++k) qo[k] = j.qoff[j.lo + k] - j.qoff[j.lo];
            j.rc = pmx_align_batch(j.cfg, m, j.qbuf + j.qoff[j.lo], qo.data(), j.rbuf + j.roff[j.lo], ro.data(), j.out + j.lo,
                                   j.stats ? j.stats + j.lo : nullptr);
        }
        if (j.rc) snprintf(j.err, sizeof j.err, "device %d: %s", j.device, pmx_last_error());
    }
};
std::mutex g_pool_mx;
std::vector<ShardWorker *> g_pool;
}  // namespace

static int multi_run(const pmx_config_t *cfg, const parasail_profile_t *profile, int64_t n,
                     const uint8_t *qbuf, const int64_t *qoff, const uint8_t *rbuf, const int64_t *roff,
                     const int *devices, int ndev, pmx_record_t *out, pmx_stats_t *stats_out)
{
    if (check_cfg(cfg)) return -1;
    if (n <= 0) return 0;
    if (!devices || ndev <= 0 || ndev > 64) { set_err("bad device list"); return -1; }
    if (!rbuf || !roff || !out || (!profile && (!qbuf || !qoff))) { set_err("null buffer"); return -1; }
    if ((cfg->want & PMX_WANT_STATS) && !stats_out) { set_err("stats requested without a stats buffer"); return -1; }
    const int have = pmx_device_count();
    for (int g = 0; g < ndev; ++g) if (devices[g] < 0 || devices[g] >= have) { set_err("device %d of the list does not exist (%d visible)", devices[g], have); return -1; }
    std::vector<int64_t> bounds((size_t)ndev + 1);
    if (pmx_shard_bounds_by_cells(n, profile ? nullptr : qoff, roff, ndev, bounds.data())) { set_err("shard planner failed"); return -1; }
    std::lock_guard<std::mutex> call_lock(g_pool_mx);          // one multi-GPU call at a time per process (the workers are shared)
    while ((int)g_pool.size() < ndev) {
        ShardWorker *w = new ShardWorker;
        try { w->th = std::thread([w] { w->loop(); }); }
        catch (const std::system_error &e) { delete w; set_err("cannot start a host thread for shard %d: %s", (int)g_pool.size(), e.what()); return -1; }
        w->th.detach();
        g_pool.push_back(w);
    }
    std::vector<ShardJob> jobs((size_t)ndev);
    for (int g = 0; g < ndev; ++g) {
        ShardJob &j = jobs[g];
        j.cfg = cfg; j.profile = profile; j.lo = bounds[g]; j.hi = bounds[g + 1];
        j.qbuf = qbuf; j.qoff = qoff; j.rbuf = rbuf; j.roff = roff; j.out = out; j.stats = stats_out; j.device = devices[g]; j.rc = 0; j.err[0] = 0;
        ShardWorker *w = g_pool[g];
        { std::lock_guard<std::mutex> lk(w->mx); w->job = &j; w->done = false; }
        w->cv.notify_all();
    }
    int rc = 0;
    for (int g = 0; g < ndev; ++g) {
        ShardWorker *w = g_pool[g];
        std::unique_lock<std::mutex> lk(w->mx);
        w->cv.wait(lk, [&] { return w->done; });
        if (jobs[g].rc && !rc) { rc = jobs[g].rc; set_err("%s", jobs[g].err); }
    }
    return rc;
}

extern "C" int pmx_align_batch_multi(const pmx_config_t *cfg, int64_t n,
                                     const uint8_t *qbuf, const int64_t *qoff, const uint8_t *rbuf, const int64_t *roff,
                                     const int *devices, int ndev, pmx_record_t *out, pmx_stats_t *stats_out)
{
    if (n > 0 && qoff && roff && (qoff[0] != 0 || roff[0] != 0)) { set_err("offset arrays must start at 0"); return -1; }
    return multi_run(cfg, nullptr, n, qbuf, qoff, rbuf, roff, devices, ndev, out, stats_out);
}

extern "C" int pmx_align_profile_batch_multi(const pmx_config_t *cfg, const parasail_profile_t *profile, int64_t n,
                                             const uint8_t *rbuf, const int64_t *roff,
                                             const int *devices, int ndev, pmx_record_t *out, pmx_stats_t *stats_out)
{
    if (!profile) { set_err("null profile"); return -1; }
    if (cfg && profile->matrix != cfg->matrix) { set_err("profile was built with a different matrix"); return -1; }
    if (n > 0 && roff && roff[0] != 0) { set_err("offset arrays must start at 0"); return -1; }
    return multi_run(cfg, profile, n, nullptr, nullptr, rbuf, roff, devices, ndev, out, stats_out);
}

extern "C" void pmx_free(void *p) { if (p && !g_text_pool.give_back(p)) free(p); }

// Page-locks a caller-owned host buffer (hipHostRegister) so that the host-buffer batch entries copy it by DMA at full PCIe rate
// instead of through the driver's pageable staging (measured: 39 -> ~55 GB/s); one-time cost, undone by pmx_host_unregister.
extern "C" int pmx_host_register(void *p, size_t bytes)
{
    if (!p || !bytes) return 0;
    HIP_OR_RET(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return 0;
}
extern "C" int pmx_host_unregister(void *p)
{
    if (!p) return 0;
    HIP_OR_RET(hipHostUnregister(p));
    return 0;
}
