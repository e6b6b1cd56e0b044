// TEST INFRASTRUCTURE (oracle): C++ restatement of kanzi-go's DivSufSort (v2/transform/DivSufSort.go, a Go port of Yuta Mori's
// libdivsufsort), function by function, so that the CPU baseline of the BWT stage times the reference's own suffix-sort
// algorithm and not a stand-in. Never linked into or called by the product library. The BWT it yields is a function of the input
// only: tests/test_oracle_units.py checks it against the oracle's SA-IS path and an independent suffix-sort definition.
// Every function cites the reference lines it follows.
#pragma once
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

namespace knzo {

class DivSufSort {
    // DivSufSort.go:19-28
    static constexpr int32_t SS_INSERTIONSORT_THRESHOLD = 8;
    static constexpr int32_t SS_BLOCKSIZE = 8192;
    static constexpr int32_t SS_MISORT_STACKSIZE = 16;
    static constexpr int32_t SS_SMERGE_STACKSIZE = 32;
    static constexpr int32_t TR_STACKSIZE = 64;
    static constexpr int32_t TR_INSERTIONSORT_THRESHOLD = 16;

    struct StackElement { int32_t a, b, c, d, e; };                // DivSufSort.go:2618-2664
    struct Stack {
        std::vector<StackElement> arr;
        int32_t index = 0;
        explicit Stack(int32_t size) : arr((size_t)size) {}
        StackElement& get(int32_t idx) { return arr[(size_t)idx]; }
        int32_t size() const { return index; }
        void push(int32_t a, int32_t b, int32_t c, int32_t d, int32_t e) {
            StackElement& elt = arr[(size_t)index];
            elt.a = a; elt.b = b; elt.c = c; elt.d = d; elt.e = e;
            index++;
        }
        StackElement* pop() {
            if (index == 0) return nullptr;
            index--;
            return &arr[(size_t)index];
        }
    };
    struct TrBudget {                                                // DivSufSort.go:2626-2632,2666-2680
        int32_t chance, remain, incVal, count;
        bool check(int32_t size) {
            if (size <= remain) { remain -= size; return true; }
            if (chance == 0) { count += size; return false; }
            remain += (incVal - size);
            chance--;
            return true;
        }
    };

    int32_t* sa = nullptr;
    const uint8_t* buffer = nullptr;
    Stack ssStack{SS_MISORT_STACKSIZE}, trStack{TR_STACKSIZE}, mergestack{SS_SMERGE_STACKSIZE};
    std::vector<int32_t> bucketA = std::vector<int32_t>(256), bucketB = std::vector<int32_t>(65536);

    static const int32_t* sqqTable() {                               // DivSufSort.go:30-47
        static const int32_t t[256] = {
            0, 16, 22, 27, 32, 35, 39, 42, 45, 48, 50, 53, 55, 57, 59, 61, 64, 65, 67, 69,
            71, 73, 75, 76, 78, 80, 81, 83, 84, 86, 87, 89, 90, 91, 93, 94, 96, 97, 98, 99,
            101, 102, 103, 104, 106, 107, 108, 109, 110, 112, 113, 114, 115, 116, 117, 118,
            119, 120, 121, 122, 123, 124, 125, 126, 128, 128, 129, 130, 131, 132, 133, 134,
            135, 136, 137, 138, 139, 140, 141, 142, 143, 144, 144, 145, 146, 147, 148, 149,
            150, 150, 151, 152, 153, 154, 155, 155, 156, 157, 158, 159, 160, 160, 161, 162,
            163, 163, 164, 165, 166, 167, 167, 168, 169, 170, 170, 171, 172, 173, 173, 174,
            175, 176, 176, 177, 178, 178, 179, 180, 181, 181, 182, 183, 183, 184, 185, 185,
            186, 187, 187, 188, 189, 189, 190, 191, 192, 192, 193, 193, 194, 195, 195, 196,
            197, 197, 198, 199, 199, 200, 201, 201, 202, 203, 203, 204, 204, 205, 206, 206,
            207, 208, 208, 209, 209, 210, 211, 211, 212, 212, 213, 214, 214, 215, 215, 216,
            217, 217, 218, 218, 219, 219, 220, 221, 221, 222, 222, 223, 224, 224, 225, 225,
            226, 226, 227, 227, 228, 229, 229, 230, 230, 231, 231, 232, 232, 233, 234, 234,
            235, 235, 236, 236, 237, 237, 238, 238, 239, 240, 240, 241, 241, 242, 242, 243,
            243, 244, 244, 245, 245, 246, 246, 247, 247, 248, 248, 249, 249, 250, 250, 251,
            251, 252, 252, 253, 253, 254, 254, 255};
        return t;
    }
    static int32_t logTable(int32_t v) {                             // _LOG_TABLE, DivSufSort.go:49-60: floor(log2(v)) for 0..255, -1 for 0
        static const struct T { int8_t t[256]; T() { for (int i = 0; i < 256; i++) { int r = -1, x = i; while (x > 0) { r++; x >>= 1; } t[i] = (int8_t)r; } } } tab;
        return tab.t[v & 0xFF];
    }
    static int32_t getIndex(int32_t a) { return a >= 0 ? a : ~a; }   // :818-824

    void reset() {                                                   // :83-89
        ssStack.index = 0; trStack.index = 0; mergestack.index = 0;
        std::fill(bucketA.begin(), bucketA.end(), 0);
        std::fill(bucketB.begin(), bucketB.end(), 0);
    }

public:
    // ComputeBWT (:179-198): bwt[] is the int32 work array (length >= n), dst receives the n BWT bytes, indexes[] the primary
    // indexes (idxCount of them); returns the primary index + 1
    int32_t computeBWT(const uint8_t* src, uint8_t* dst, int32_t* bwt, int32_t length, uint32_t* indexes, int32_t idxCount) {
        buffer = src;
        sa = bwt;
        reset();
        const int32_t m = sortTypeBstar(length);
        const int32_t pIdx = constructBWT(length, m, indexes, idxCount);
        dst[0] = src[length - 1];
        for (int32_t i = 0; i < pIdx; i++) dst[i + 1] = (uint8_t)bwt[i];
        for (int32_t i = pIdx + 1; i < length; i++) dst[i] = (uint8_t)bwt[i];
        return pIdx + 1;
    }
    // ComputeSuffixArray (:93-99)
    void computeSuffixArray(const uint8_t* src, int32_t* saOut, int32_t length) {
        buffer = src;
        sa = saOut;
        reset();
        const int32_t m = sortTypeBstar(length);
        constructSuffixArray(length, m);
    }

private:
    void constructSuffixArray(int32_t n, int32_t m) {                // :101-176
        if (m > 0) {
            for (int c1 = 254; c1 >= 0; c1--) {
                const int idx = c1 << 8;
                const int32_t i = bucketB[idx + c1 + 1];
                int32_t k = 0;
                int c2 = -1;
                for (int32_t j = bucketA[c1 + 1] - 1; j >= i; j--) {   // scan from right to left
                    int32_t s = sa[j];
                    sa[j] = ~s;
                    if (s <= 0) continue;
                    s--;
                    const int c0 = buffer[s];
                    if (s > 0 && (int)buffer[s - 1] > c0) s = ~s;
                    if (c0 != c2) {
                        if (c2 >= 0) bucketB[idx + c2] = k;
                        c2 = c0;
                        k = bucketB[idx + c2];
                    }
                    sa[k] = s;
                    k--;
                }
            }
        }
        int c2 = buffer[n - 1];
        int32_t k = bucketA[c2];
        if ((int)buffer[n - 2] < c2) sa[k] = ~(n - 1); else sa[k] = n - 1;
        k++;
        for (int32_t i = 0; i < n; i++) {                             // scan from left to right
            int32_t s = sa[i];
            if (s <= 0) { sa[i] = ~s; continue; }
            s--;
            const int c0 = buffer[s];
            if (s == 0 || (int)buffer[s - 1] < c0) s = ~s;
            if (c0 != c2) { bucketA[c2] = k; c2 = c0; k = bucketA[c2]; }
            sa[k] = s;
            k++;
        }
    }

    int32_t constructBWT(int32_t n, int32_t m, uint32_t* indexes, int32_t idxCount) {   // :200-311
        int32_t pIdx = -1;
        int32_t step = n / idxCount;
        if (step * idxCount != n) step++;
        if (m > 0) {
            for (int c1 = 254; c1 >= 0; c1--) {
                const int idx = c1 << 8;
                const int32_t i = bucketB[idx + c1 + 1];
                int32_t k = 0;
                int c2 = -1;
                for (int32_t j = bucketA[c1 + 1] - 1; j >= i; j--) {   // scan from right to left
                    int32_t s = sa[j];
                    if (s <= 0) {
                        if (s != 0) sa[j] = ~s;
                        continue;
                    }
                    if (s % step == 0) indexes[s / step] = (uint32_t)(j + 1);
                    s--;
                    const int c0 = buffer[s];
                    sa[j] = ~(int32_t)c0;
                    if (s > 0 && (int)buffer[s - 1] > c0) s = ~s;
                    if (c0 != c2) {
                        if (c2 >= 0) bucketB[idx + c2] = k;
                        c2 = c0;
                        k = bucketB[idx + c2];
                    }
                    sa[k] = s;
                    k--;
                }
            }
        }
        uint8_t c2 = buffer[n - 1];
        int32_t k = bucketA[c2];
        if (buffer[n - 2] < c2) {
            if ((n - 1) % step == 0) indexes[(n - 1) / step] = (uint32_t)n;
            sa[k] = ~(int32_t)buffer[n - 2];
        } else sa[k] = n - 1;
        k++;
        for (int32_t i = 0; i < n; i++) {                             // scan from left to right
            int32_t s = sa[i];
            if (s <= 0) {
                if (s != 0) sa[i] = ~s; else pIdx = i;
                continue;
            }
            if ((s % step) == 0) indexes[s / step] = (uint32_t)(i + 1);
            s--;
            const uint8_t c0 = buffer[s];
            sa[i] = (int32_t)c0;
            if (c0 != c2) { bucketA[c2] = k; c2 = c0; k = bucketA[c2]; }
            if (s > 0 && buffer[s - 1] < c0) {
                if ((s % step) == 0) indexes[s / step] = (uint32_t)(k + 1);
                s = ~(int32_t)buffer[s - 1];
            }
            sa[k] = s;
            k++;
        }
        indexes[0] = (uint32_t)(pIdx + 1);
        return pIdx;
    }

    int32_t sortTypeBstar(int32_t n) {                               // :313-525
        int32_t m = n;
        uint8_t c0 = buffer[n - 1];
        int32_t* arr = sa;
        // count the occurrences of the first one or two characters of each type A, B and B* suffix; store the start of every B* suffix
        for (int32_t i = n - 1; i >= 0;) {
            uint8_t c1 = c0;
            while (c0 >= c1) {
                c1 = c0;
                bucketA[c1]++;
                i--;
                if (i < 0) break;
                c0 = buffer[i];
            }
            if (i < 0) break;
            bucketB[((int)c0 << 8) + (int)c1]++;
            m--;
            arr[m] = i;
            i--;
            c1 = c0;
            while (i >= 0) {
                c0 = buffer[i];
                if (c0 > c1) break;
                bucketB[((int)c1 << 8) + (int)c0]++;
                c1 = c0;
                i--;
            }
        }
        m = n - m;
        int x0 = 0;
        // start/end point of each bucket
        for (int32_t i = 0, j = 0; x0 < 256; x0++) {
            const int32_t t = i + bucketA[x0];
            bucketA[x0] = i + j;                                     // start point
            const int idx = x0 << 8;
            i = t + bucketB[idx + x0];
            for (int x1 = x0 + 1; x1 < 256; x1++) {
                j += bucketB[idx + x1];
                bucketB[idx + x1] = j;                               // end point
                i += bucketB[(x1 << 8) + x0];
            }
        }
        if (m > 0) {
            // sort the type B* suffixes by their first two characters
            const int32_t pab = n - m;
            for (int32_t i = m - 2; i >= 0; i--) {
                const int32_t t = arr[pab + i];
                const int idx = ((int)buffer[t] << 8) + (int)buffer[t + 1];
                bucketB[idx]--;
                arr[bucketB[idx]] = i;
            }
            const int32_t t = arr[pab + m - 1];
            const int c3 = ((int)buffer[t] << 8) + (int)buffer[t + 1];
            bucketB[c3]--;
            arr[bucketB[c3]] = m - 1;
            // sort the type B* substrings using ssSort
            const int32_t bufSize = n - m - m;
            x0 = 254;
            for (int32_t j = m; j > 0; x0--) {
                const int idx = x0 << 8;
                for (int x1 = 255; x1 > x0; x1--) {
                    const int32_t i = bucketB[idx + x1];
                    if (j - i > 1) ssSort(pab, i, j, m, bufSize, 2, n, arr[i] == m - 1);
                    j = i;
                }
            }
            // ranks of type B* substrings
            for (int32_t i = m - 1; i >= 0; i--) {
                if (arr[i] >= 0) {
                    const int32_t j = i;
                    for (;;) {
                        arr[m + arr[i]] = i;
                        i--;
                        if (i < 0 || arr[i] < 0) break;
                    }
                    arr[i + 1] = i - j;
                    if (i <= 0) break;
                }
                const int32_t j = i;
                for (;;) {
                    arr[i] = ~arr[i];
                    arr[m + arr[i]] = j;
                    i--;
                    if (arr[i] >= 0) break;
                }
                arr[m + arr[i]] = j;
            }
            // inverse suffix array of the type B* suffixes
            trSort(m, 1);
            // set the sorted order of type B* suffixes
            c0 = buffer[n - 1];
            uint8_t c1;
            for (int32_t i = n - 1, j = m; i >= 0;) {
                i--;
                c1 = c0;
                while (i >= 0) {
                    c0 = buffer[i];
                    if (c0 < c1) break;
                    c1 = c0;
                    i--;
                }
                if (i >= 0) {
                    const int32_t tt = i;
                    i--;
                    c1 = c0;
                    while (i >= 0) {
                        c0 = buffer[i];
                        if (c0 > c1) break;
                        c1 = c0;
                        i--;
                    }
                    j--;
                    if (tt == 0 || tt - i > 1) arr[arr[m + j]] = tt; else arr[arr[m + j]] = ~tt;
                }
            }
            // start/end point of each bucket
            bucketB[65535] = n;                                      // end
            int32_t k = m - 1;
            for (x0 = 254; x0 >= 0; x0--) {
                int32_t i = bucketA[x0 + 1] - 1;
                const int x2 = x0 << 8;
                for (int x1 = 255; x1 > x0; x1--) {
                    const int32_t tt = i - bucketB[(x1 << 8) + x0];
                    bucketB[(x1 << 8) + x0] = i;                     // end point
                    i = tt;
                    // move all type B* suffixes to the correct position (typically very few copies)
                    for (int32_t j = bucketB[x2 + x1]; j <= k;) {
                        arr[i] = arr[k];
                        i--;
                        k--;
                    }
                }
                bucketB[x2 + x0 + 1] = i - bucketB[x2 + x0] + 1;     // start point
                bucketB[x2 + x0] = i;                                // end point
            }
        }
        return m;
    }

    // ---- sub string sort ------------------------------------------------------------------------------------------------
    void ssSort(int32_t pa, int32_t first, int32_t last, int32_t buf, int32_t bufSize, int32_t depth, int32_t n, bool lastSuffix) {   // :528-607
        if (lastSuffix) first++;
        int32_t limit = 0;
        int32_t middle = last;
        if (bufSize < SS_BLOCKSIZE && bufSize < last - first) {
            limit = ssIsqrt(last - first);
            if (bufSize < limit) {
                if (limit > SS_BLOCKSIZE) limit = SS_BLOCKSIZE;
                middle = last - limit;
                buf = middle;
                bufSize = limit;
            } else limit = 0;
        }
        int32_t a;
        int32_t i = 0;
        for (a = first; middle - a > SS_BLOCKSIZE; a += SS_BLOCKSIZE) {
            ssMultiKeyIntroSort(pa, a, a + SS_BLOCKSIZE, depth);
            int32_t curBufSize = last - (a + SS_BLOCKSIZE);
            int32_t curBuf;
            if (curBufSize > bufSize) curBuf = a + SS_BLOCKSIZE;
            else { curBufSize = bufSize; curBuf = buf; }
            int32_t k = SS_BLOCKSIZE;
            int32_t b = a;
            for (int32_t j = i; (j & 1) != 0; j >>= 1) {
                ssSwapMerge(pa, b - k, b, b + k, curBuf, curBufSize, depth);
                b -= k;
                k <<= 1;
            }
            i++;
        }
        ssMultiKeyIntroSort(pa, a, middle, depth);
        int32_t k = SS_BLOCKSIZE;
        while (i != 0) {
            if ((i & 1) != 0) {
                ssSwapMerge(pa, a - k, a, middle, buf, bufSize, depth);
                a -= k;
            }
            k <<= 1;
            i >>= 1;
        }
        if (limit != 0) {
            ssMultiKeyIntroSort(pa, middle, last, depth);
            ssInplaceMerge(pa, first, middle, last, depth);
        }
        if (lastSuffix) {
            i = sa[first - 1];
            const int32_t p1 = sa[pa + i];
            for (a = first; a < last && (sa[a] < 0 || ssCompare4(p1, n - 2, pa + sa[a], depth) > 0); a++) sa[a - 1] = sa[a];
            sa[a - 1] = i;
        }
    }

    int ssCompare4(int32_t pa, int32_t pb, int32_t p2, int32_t depth) const {   // :609-640
        const int32_t u1n = pb + 2;
        int32_t u1 = pa + depth;
        const int32_t u2n = sa[p2 + 1] + 2;
        int32_t u2 = sa[p2] + depth;
        if (u1n - u1 > u2n - u2) { while (u2 < u2n && buffer[u1] == buffer[u2]) { u1++; u2++; } }
        else { while (u1 < u1n && buffer[u1] == buffer[u2]) { u1++; u2++; } }
        if (u1 < u1n) {
            if (u2 < u2n) return (int)buffer[u1] - (int)buffer[u2];
            return 1;
        }
        if (u2 < u2n) return -1;
        return 0;
    }

    int ssCompare3(int32_t p1, int32_t p2, int32_t depth) const {     // :642-674
        const int32_t u1n = sa[p1 + 1] + 2;
        int32_t u1 = sa[p1] + depth;
        const int32_t u2n = sa[p2 + 1] + 2;
        int32_t u2 = sa[p2] + depth;
        const uint8_t* buf = buffer;
        if (u1n - u1 > u2n - u2) { while (u2 < u2n && buf[u1] == buf[u2]) { u1++; u2++; } }
        else { while (u1 < u1n && buf[u1] == buf[u2]) { u1++; u2++; } }
        if (u1 < u1n) {
            if (u2 < u2n) return (int)buf[u1] - (int)buf[u2];
            return 1;
        }
        if (u2 < u2n) return -1;
        return 0;
    }

    void ssInplaceMerge(int32_t pa, int32_t first, int32_t middle, int32_t last, int32_t depth) {   // :676-742
        int32_t* arr = sa;
        for (;;) {
            int32_t p, x;
            if (arr[last - 1] < 0) { x = 1; p = pa + ~arr[last - 1]; }
            else { x = 0; p = pa + arr[last - 1]; }
            int32_t a = first;
            int r = -1;
            int32_t half = (middle - first) >> 1;
            for (int32_t length = middle - first; length > 0; half >>= 1) {
                const int32_t b = a + half;
                const int32_t c = arr[b] >= 0 ? arr[b] : ~arr[b];
                const int q = ssCompare3(pa + c, p, depth);
                if (q >= 0) r = q;
                else { a = b + 1; half -= ((length & 1) ^ 1); }
                length = half;
            }
            if (a < middle) {
                if (r == 0) arr[a] = ~arr[a];
                ssRotate(a, middle, last);
                last -= (middle - a);
                middle = a;
                if (first == middle) break;
            }
            last--;
            if (x != 0) {
                last--;
                while (arr[last] < 0) last--;
            }
            if (middle == last) break;
        }
    }

    void ssRotate(int32_t first, int32_t middle, int32_t last) {      // :744-807
        int32_t l = middle - first;
        int32_t r = last - middle;
        int32_t* arr = sa;
        while (l > 0 && r > 0) {
            if (l == r) { ssBlockSwap(first, middle, l); break; }
            if (l < r) {
                int32_t a = last - 1;
                int32_t b = middle - 1;
                int32_t t = arr[a];
                for (;;) {
                    arr[a] = arr[b]; a--;
                    arr[b] = arr[a]; b--;
                    if (b < first) {
                        arr[a] = t;
                        last = a;
                        r -= (l + 1);
                        if (r <= l) break;
                        a--;
                        b = middle - 1;
                        t = arr[a];
                    }
                }
            } else {
                int32_t a = first;
                int32_t b = middle;
                int32_t t = arr[a];
                for (;;) {
                    arr[a] = arr[b]; a++;
                    arr[b] = arr[a]; b++;
                    if (last <= b) {
                        arr[a] = t;
                        first = a + 1;
                        l -= (r + 1);
                        if (l <= r) break;
                        a++;
                        b = middle;
                        t = arr[a];
                    }
                }
            }
        }
    }

    void ssBlockSwap(int32_t a, int32_t b, int32_t n) {               // :809-816
        while (n > 0) { std::swap(sa[a], sa[b]); n--; a++; b++; }
    }

#include "divsufsort_ss.inc"
#include "divsufsort_tr.inc"
};

}  // namespace knzo
