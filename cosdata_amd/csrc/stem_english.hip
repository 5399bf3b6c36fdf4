// stem_english.hip — the English Snowball stemmer ("Porter2"), host code, for process_text (indexes/tf_idf/mod.rs:317-340:
// `Stemmer::create()` "Create an English stemmer", then `stemmer.stem(&lower)`).
//
// The reference's stemmer is an un-vendored git dependency (snowball-stemmer 0.1.0,
// git+https://github.com/cosdata/snowball-stemmer.git#dcbd7da7b3ad86cd8ba240b91747dadb58aa77f9, Cargo.lock:2571-2573): its source is
// not in /root/reference, so this file restates the PUBLISHED algorithm that crate ports — the Snowball project's `english`
// stemmer (prelude / mark_regions / Step_1a .. Step_5 / exception1 / exception2 / postlude), in the long-stable form of the
// definition: regions R1 / R2 with the gener- / commun- / arsen- prefixes, short syllables, the -li endings cdeghkmnrt.
// Pinned by the sample vocabulary of the algorithm's own description and by an independent Python restatement
// (tests/test_stem_english.py); identity with the fork at that commit is unpinned.
// Works on Unicode scalars: `hop`, `next` and the vowel / non-vowel classes count characters, not bytes.
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace {

typedef std::vector<uint32_t> W; // the word as code points; 'Y' (0x59) marks a consonantal y

inline bool is_v(uint32_t c) { return c == 'a' || c == 'e' || c == 'i' || c == 'o' || c == 'u' || c == 'y'; }
inline bool is_v_wxy(uint32_t c) { return is_v(c) || c == 'w' || c == 'x' || c == 'Y'; }
inline bool valid_li(uint32_t c) { return c == 'c' || c == 'd' || c == 'e' || c == 'g' || c == 'h' || c == 'k' || c == 'm' || c == 'n' || c == 'r' || c == 't'; }

inline bool ends(const W &w, const char *s, size_t upto) { // w[0..upto) ends with s
    const size_t n = strlen(s);
    if (upto < n) return false;
    for (size_t i = 0; i < n; i++)
        if (w[upto - n + i] != (unsigned char)s[i]) return false;
    return true;
}
inline bool ends(const W &w, const char *s) { return ends(w, s, w.size()); }
inline bool equals(const W &w, const char *s) { return w.size() == strlen(s) && ends(w, s); }
inline void set_suffix(W &w, size_t cut, const char *repl) { // drop the last `cut` characters, append repl
    w.resize(w.size() - cut);
    for (const char *p = repl; *p; p++) w.push_back((unsigned char)*p);
}
// longest suffix of `list` (NULL-terminated) that w ends with; -1 if none
inline int longest(const W &w, const char *const *list) {
    int best = -1;
    size_t bl = 0;
    for (int i = 0; list[i]; i++) {
        const size_t n = strlen(list[i]);
        if ((best < 0 || n > bl) && ends(w, list[i])) { best = i; bl = n; }
    }
    return best;
}
inline bool has_vowel(const W &w, size_t upto) { // `gopast v` backwards from position upto: a vowel somewhere in w[0..upto)
    for (size_t i = 0; i < upto; i++)
        if (is_v(w[i])) return true;
    return false;
}
// shortv with the cursor at `pos` (backward mode): w[0..pos) ends in a short syllable
inline bool shortv(const W &w, size_t pos) {
    if (pos >= 3 && !is_v_wxy(w[pos - 1]) && is_v(w[pos - 2]) && !is_v(w[pos - 3])) return true;
    if (pos == 2 && !is_v(w[1]) && is_v(w[0])) return true;
    return false;
}

void stem_word(W &w) {
    // exception1: whole-word special forms
    static const char *const EX1[][2] = {{"skis", "ski"}, {"skies", "sky"}, {"dying", "die"}, {"lying", "lie"}, {"tying", "tie"}, {"idly", "idl"},
                                         {"gently", "gentl"}, {"ugly", "ugli"}, {"early", "earli"}, {"only", "onli"}, {"singly", "singl"},
                                         {"sky", "sky"}, {"news", "news"}, {"howe", "howe"}, {"atlas", "atlas"}, {"cosmos", "cosmos"},
                                         {"bias", "bias"}, {"andes", "andes"}};
    for (const auto &e : EX1)
        if (equals(w, e[0])) { set_suffix(w, w.size(), e[1]); return; }
    if (w.size() < 3) return; // not hop 3

    // prelude: initial apostrophe off; initial y and every y after a vowel become Y
    bool y_found = false;
    if (w[0] == '\'') w.erase(w.begin());
    if (!w.empty() && w[0] == 'y') { w[0] = 'Y'; y_found = true; }
    for (size_t i = 1; i < w.size(); i++)
        if (w[i] == 'y' && is_v(w[i - 1])) { w[i] = 'Y'; y_found = true; }

    // mark_regions
    size_t p1 = w.size(), p2 = w.size();
    {
        size_t c = 0;
        bool ok = true;
        static const char *const PRE[] = {"gener", "commun", "arsen"};
        bool pre = false;
        for (const char *p : PRE) {
            const size_t n = strlen(p);
            if (w.size() >= n) {
                bool m = true;
                for (size_t i = 0; i < n; i++) m &= w[i] == (unsigned char)p[i];
                if (m) { c = n; pre = true; break; }
            }
        }
        auto gopast = [&](bool vowel) { // move past the next character of the class; false at the limit
            while (c < w.size() && is_v(w[c]) != vowel) c++;
            if (c >= w.size()) return false;
            c++;
            return true;
        };
        if (!pre) ok = gopast(true) && gopast(false);
        if (ok) {
            p1 = c;
            if (gopast(true) && gopast(false)) p2 = c;
        }
    }

    // Step_1a (first the apostrophe forms, "step 0" of the description)
    {
        static const char *const AP[] = {"'s'", "'s", "'", nullptr};
        const int a = longest(w, AP);
        if (a >= 0) w.resize(w.size() - strlen(AP[a]));
        static const char *const S1A[] = {"sses", "ied", "ies", "s", "us", "ss", nullptr};
        const int k = longest(w, S1A);
        if (k == 0) set_suffix(w, 4, "ss");
        else if (k == 1 || k == 2) set_suffix(w, 3, w.size() - 3 >= 2 ? "i" : "ie"); // (hop 2 <-'i') or <-'ie'
        else if (k == 3) { // next gopast v delete: a vowel before the letter that precedes the s
            if (w.size() >= 2 && has_vowel(w, w.size() - 2)) w.resize(w.size() - 1);
        }
    }
    // exception2: invariant after step 1a
    static const char *const EX2[] = {"inning", "outing", "canning", "herring", "earring", "proceed", "exceed", "succeed"};
    bool invariant = false;
    for (const char *e : EX2) invariant |= equals(w, e);
    if (!invariant) {
        { // Step_1b
            static const char *const S1B[] = {"eed", "eedly", "ed", "edly", "ing", "ingly", nullptr};
            const int k = longest(w, S1B);
            if (k == 0 || k == 1) {
                const size_t n = strlen(S1B[k]);
                if (p1 <= w.size() - n) set_suffix(w, n, "ee");
            } else if (k >= 2) {
                const size_t n = strlen(S1B[k]);
                if (has_vowel(w, w.size() - n)) {
                    w.resize(w.size() - n);
                    static const char *const T[] = {"at", "bl", "iz", "bb", "dd", "ff", "gg", "mm", "nn", "pp", "rr", "tt", nullptr};
                    const int t = longest(w, T);
                    if (t >= 0 && t <= 2) w.push_back('e');
                    else if (t >= 3) w.pop_back();
                    else if (w.size() == p1 && shortv(w, w.size())) w.push_back('e'); // atmark p1 test shortv
                }
            }
        }
        // Step_1c: y / Y after a non-vowel that is not the first letter
        if (w.size() >= 3 && (w.back() == 'y' || w.back() == 'Y') && !is_v(w[w.size() - 2])) w.back() = 'i';
        { // Step_2: longest suffix, must lie in R1
            static const char *const S[] = {"tional", "enci", "anci", "abli", "entli", "izer", "ization", "ational", "ation", "ator", "alism", "aliti",
                                            "alli", "fulness", "ousli", "ousness", "iveness", "iviti", "biliti", "bli", "ogi", "fulli", "lessli", "li", nullptr};
            static const char *const R[] = {"tion", "ence", "ance", "able", "ent", "ize", "ize", "ate", "ate", "ate", "al", "al",
                                            "al", "ful", "ous", "ous", "ive", "ive", "ble", "ble", "og", "ful", "less", ""};
            const int k = longest(w, S);
            if (k >= 0) {
                const size_t n = strlen(S[k]);
                if (p1 <= w.size() - n) {
                    if (k == 20) { if (w.size() > n && w[w.size() - n - 1] == 'l') set_suffix(w, n, R[k]); }
                    else if (k == 23) { if (w.size() > n && valid_li(w[w.size() - n - 1])) set_suffix(w, n, ""); }
                    else set_suffix(w, n, R[k]);
                }
            }
        }
        { // Step_3: in R1 (ative: in R2)
            static const char *const S[] = {"tional", "ational", "alize", "icate", "iciti", "ical", "ful", "ness", "ative", nullptr};
            static const char *const R[] = {"tion", "ate", "al", "ic", "ic", "ic", "", "", ""};
            const int k = longest(w, S);
            if (k >= 0) {
                const size_t n = strlen(S[k]);
                if (p1 <= w.size() - n && (k != 8 || p2 <= w.size() - n)) set_suffix(w, n, R[k]);
            }
        }
        { // Step_4: in R2, delete (ion: after s or t)
            static const char *const S[] = {"al", "ance", "ence", "er", "ic", "able", "ible", "ant", "ement", "ment", "ent", "ism", "ate", "iti", "ous",
                                            "ive", "ize", "ion", nullptr};
            const int k = longest(w, S);
            if (k >= 0) {
                const size_t n = strlen(S[k]);
                if (p2 <= w.size() - n) {
                    if (k != 17) w.resize(w.size() - n);
                    else if (w.size() > n && (w[w.size() - n - 1] == 's' || w[w.size() - n - 1] == 't')) w.resize(w.size() - n);
                }
            }
        }
        // Step_5
        if (!w.empty() && w.back() == 'e') {
            const size_t c = w.size() - 1;
            if (p2 <= c || (p1 <= c && !shortv(w, c))) w.pop_back();
        } else if (!w.empty() && w.back() == 'l') {
            const size_t c = w.size() - 1;
            if (p2 <= c && c >= 1 && w[c - 1] == 'l') w.pop_back();
        }
    }
    if (y_found)
        for (auto &c : w)
            if (c == 'Y') c = 'y';
}

} // namespace

// cos_stem_fn-compatible entry point (include/cosdata_hip.h): token (UTF-8, already lowercased by process_text) -> stem.
// Returns the number of bytes the stem needs; writes min(that, out_cap) of them.
extern "C" size_t cos_stem_english(void *, const char *token, size_t token_len, char *out, size_t out_cap) {
    W w;
    w.reserve(token_len);
    const unsigned char *s = (const unsigned char *)token;
    for (size_t i = 0; i < token_len;) { // UTF-8 -> scalars (invalid bytes pass through one by one)
        const unsigned char c = s[i];
        int extra = c < 0x80 ? 0 : (c >= 0xF0 ? 3 : (c >= 0xE0 ? 2 : (c >= 0xC0 ? 1 : -1)));
        uint32_t cp = c;
        bool ok = extra >= 0 && i + (size_t)extra < token_len + (extra == 0 ? 1 : 0) && (extra == 0 || i + (size_t)extra < token_len);
        if (ok && extra > 0) {
            cp = c & (0x3F >> extra);
            for (int k = 1; k <= extra; k++) {
                if ((s[i + k] & 0xC0) != 0x80) { ok = false; break; }
                cp = (cp << 6) | (s[i + k] & 0x3F);
            }
        }
        if (!ok || extra < 0) { w.push_back(0x110000u + c); i++; continue; } // opaque non-vowel, restored byte for byte below
        w.push_back(cp);
        i += (size_t)extra + 1;
    }
    stem_word(w);
    std::string o;
    for (uint32_t cp : w) {
        if (cp >= 0x110000u) o.push_back((char)(cp - 0x110000u));
        else if (cp < 0x80) o.push_back((char)cp);
        else if (cp < 0x800) { o.push_back((char)(0xC0 | (cp >> 6))); o.push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) { o.push_back((char)(0xE0 | (cp >> 12))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
        else { o.push_back((char)(0xF0 | (cp >> 18))); o.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
    }
    if (out && out_cap) memcpy(out, o.data(), o.size() < out_cap ? o.size() : out_cap);
    return o.size();
}
