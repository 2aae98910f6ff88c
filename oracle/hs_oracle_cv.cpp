// ORACLE (test infrastructure, CPU only): restatement of stage 3, HS_call_variants.
// See hs_oracle.h for the rules. Citations are to /root/reference/src/<file>:<line>.
#include "hs_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>

namespace hso {

static inline int acgt_index(char c) {  // string("ACGT-").find(c), call_variants.cpp:62,238
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; case '-': return 4; }
    std::fprintf(stderr, "oracle: base outside ACGT- reached the pileup (reference behaviour undefined)\n");
    std::exit(2);
}

// tools.cpp:27-57
std::string convert_cigar(const std::string& cigar) {
    if (cigar == "*") return "";
    std::string res, num;
    for (char c : cigar) {
        if (c >= '0' && c <= '9') num += c;
        else {
            int n = 0;
            try { n = std::stoi(num); } catch (...) {
                std::cout << "ERROR : could not convert " << cigar << " to int" << std::endl;
                std::exit(1);
            }
            res.append((size_t)std::max(n, 0), c);
            num = "";
        }
    }
    return res;
}

// sequence.cpp:13-52: construct (2-bit) then str()
std::string two_bit_filter(const std::string& s) {
    std::string o(s.size(), 'T');
    for (size_t i = 0; i < s.size(); i++) {
        char c = s[i];
        o[i] = (c == 'A' || c == 'C' || c == 'G') ? c : 'T';
    }
    return o;
}

// sequence.cpp:54-65
std::string reverse_complement(const std::string& s) {
    std::string o(s.size(), 'N');
    for (size_t i = 0; i < s.size(); i++) {
        char c = s[s.size() - 1 - i];
        o[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
    }
    return o;
}

// call_variants.cpp:50-437 (SAM branch :105-116, CIGAR walk :215-352, newref :366-376)
MsaResult generate_msa(const Contig& c, const std::vector<std::string>& read_seq) {
    MsaResult R;
    const std::string& consensus = c.seq;
    const size_t L = consensus.size();
    R.cols.assign(L, Column());
    for (size_t i = 0; i < L; i++) R.cols[i].pos = (int)i;

    float totalDistance = 0;               // :67 (a *float* counter)
    double totalLengthOfAlignment = 1;     // :68

    for (size_t n = 0; n < c.recs.size(); n++) {
        const Record& rec = c.recs[n];
        std::string read = rec.strand ? read_seq[rec.read] : reverse_complement(read_seq[rec.read]);
        std::string alignment = convert_cigar(rec.cigar);
        int indexQuery = rec.position_2_1;   // :189 (consensus coordinate)
        int indexTarget = 0;                 // :190 (read coordinate)
        char ppp = 'A', pp = 'C', p = 'G';   // :212-214
        long nerr = 0, nlen = 0;
        for (size_t l = 0; l < alignment.size(); l++) {
            if (indexQuery >= 0 && (size_t)indexQuery < L) {   // :217 (int vs size_t: negative is "false")
                char a = alignment[l];
                if (a == '=' || a == 'X' || a == 'M') {        // :226-268
                    if ((size_t)indexTarget >= read.size()) {
                        std::fprintf(stderr, "oracle: CIGAR runs past the end of read %s\n", rec.read_name.c_str());
                        std::exit(2);
                    }
                    ppp = pp; pp = p; p = read[indexTarget];
                    unsigned char three_mer = (unsigned char)('!' + 5 * acgt_index(ppp) + acgt_index(pp) + 25 * acgt_index(p));
                    R.cols[indexQuery].readIdxs.push_back((unsigned)n);
                    R.cols[indexQuery].content.push_back(three_mer);
                    if (read[indexTarget] != consensus[indexQuery]) { totalDistance += 1; nerr++; }
                    totalLengthOfAlignment += 1; nlen++;
                    indexQuery++; indexTarget++;
                } else if (a == 'S' || a == 'H') {             // :269-273
                    indexTarget++;
                } else if (a == 'D') {                         // :274-310
                    ppp = pp; pp = p; p = '-';
                    unsigned char three_mer = (unsigned char)('!' + 5 * acgt_index(ppp) + acgt_index(pp) + 25 * acgt_index(p));
                    R.cols[indexQuery].readIdxs.push_back((unsigned)n);
                    R.cols[indexQuery].content.push_back(three_mer);
                    indexQuery++;
                    totalDistance += 1; nerr++;
                    totalLengthOfAlignment += 1; nlen++;
                } else if (a == 'I') {                         // :311-342
                    if ((size_t)indexTarget >= read.size()) {
                        std::fprintf(stderr, "oracle: CIGAR runs past the end of read %s\n", rec.read_name.c_str());
                        std::exit(2);
                    }
                    ppp = pp; pp = p; p = read[indexTarget];
                    indexTarget++;
                    totalDistance += 1; nerr++;
                    totalLengthOfAlignment += 1; nlen++;
                }
            }
        }
        R.q_end.push_back(indexQuery);       // :354
        R.n_err.push_back(nerr);
        R.n_len.push_back(nlen);
    }

    // :366-376
    unsigned char a = 'A', b = 'C', d = 'G';
    R.newref.clear();
    for (char i : consensus) {
        a = b; b = d; d = (unsigned char)i;
        R.newref += (char)(unsigned char)('!' + 5 * acgt_index((char)a) + acgt_index((char)b) + 25 * acgt_index((char)d));
    }
    R.meanDistance = (float)(totalDistance / totalLengthOfAlignment);   // :434 (float / double -> double -> float)
    return R;
}

// call_variants.cpp:447-567
CallResult call_variants(std::vector<Column>& snps, const std::string& ref, float meanError,
                         float automatic_snp_threshold) {
    CallResult R;
    int minimumNumberOfReadsToBeConsideredSuspect = 5;
    if (meanError < 0.015) minimumNumberOfReadsToBeConsideredSuspect = 3;   // :464 (float vs double literal)

    double depthOfCoverage = 0;
    int posoflastsnp = -5;
    const size_t L = ref.size();
    R.k0.assign(L, 0); R.k1.assign(L, 0); R.c0.assign(L, 0); R.c1.assign(L, 0); R.c2.assign(L, 0);
    for (int position = 0; (size_t)position < L; position++) {
        RHMap<unsigned char, int> content;                                  // :477
        for (size_t n = 0; n < snps[position].content.size(); n++) {        // :479 (short counter in the reference)
            unsigned char base = snps[position].content[n];
            if (!content.contains(base)) content[base] = 0;
            if (base != ' ') { content[base] += 1; depthOfCoverage += 1; }
        }
        content[(unsigned char)0] = 0;                                      // :492-494
        content[(unsigned char)1] = 0;
        content[(unsigned char)2] = 0;

        std::vector<std::pair<unsigned char, int>> content_sorted;          // :497-501
        content.for_each([&](unsigned char k, int v) { content_sorted.push_back(std::make_pair(k, v)); });
        std::sort(content_sorted.begin(), content_sorted.end(),
                  [](const std::pair<unsigned char, int>& a, const std::pair<unsigned char, int>& b) { return a.second > b.second; });

        if (content_sorted.size() > 1) {                                    // :503-507
            snps[position].ref_base = content_sorted[0].first;
            snps[position].second_base = content_sorted[1].first;
            snps[position].pos = position;
        }
        R.k0[position] = content_sorted[0].first; R.k1[position] = content_sorted[1].first;
        R.c0[position] = content_sorted[0].second; R.c1[position] = content_sorted[1].second;
        R.c2[position] = content_sorted[2].second;

        const int k0 = content_sorted[0].first, k1 = content_sorted[1].first;   // unsigned char promoted to int
        if (content_sorted[1].second > minimumNumberOfReadsToBeConsideredSuspect
            && (content_sorted[1].second > content_sorted[2].second * 5 || minimumNumberOfReadsToBeConsideredSuspect == 2)
            && k0 % 5 != k1 % 5
            && ((k1 - '!') % 5 != 4 || (k1 / 5 % 5 != k0 % 5 && k1 / 25 % 5 != k0 % 5))
            && position - posoflastsnp > 5) {                               // :525-529
            if ((float)content_sorted[1].second > automatic_snp_threshold * (float)content_sorted[0].second)   // :531
                R.automatic.push_back(snps[position]);
            posoflastsnp = position;
            Column snp;                                                     // :548-560
            snp.pos = position;
            snp.ref_base = (unsigned char)(char)content_sorted[0].first;
            snp.second_base = (unsigned char)(char)content_sorted[1].first;
            snp.readIdxs = snps[position].readIdxs;
            snp.content = snps[position].content;
            R.suspicious.push_back(snp);
        }
    }
    R.depth = (float)(depthOfCoverage / (double)L);                         // :565
    return R;
}

// ------------------------------------------------------------------------------------------------
// Partition (Partition.cpp)
// ------------------------------------------------------------------------------------------------
Partition::Partition(const Column& snp, int pos, unsigned char ref_base) {   // :32-83
    pos_left = pos; pos_right = pos; conf_score = 0; number_of_correlating_snps = 0;
    readIdx.assign(snp.readIdxs.begin(), snp.readIdxs.end());
    RHMap<unsigned char, int> content;
    for (size_t c = 0; c < snp.content.size(); c++) {
        if (!content.contains(snp.content[c])) content[snp.content[c]] = 1;
        else content[snp.content[c]] += 1;
    }
    unsigned char mostFrequent = ref_base;
    (void)content[ref_base];                                                 // :55 inserts ref_base if absent
    unsigned char secondFrequent = 0;   // uninitialised in the reference when the column is mono-allelic
    int maxFrequence2 = -1;
    content.for_each([&](unsigned char k, int v) {
        if (ref_base != k) { if (v > maxFrequence2) { secondFrequent = k; maxFrequence2 = v; } }
    });
    for (size_t i = 0; i < snp.content.size(); i++) {
        if (snp.content[i] == mostFrequent) mostFrequentBases.push_back(1);
        else if (snp.content[i] == secondFrequent) mostFrequentBases.push_back(-1);
        else mostFrequentBases.push_back(0);
        lessFrequence.push_back(0);
        moreFrequence.push_back(1);
    }
    numberOfOccurences = 1;
}

bool Partition::isInformative(bool lastReadBiased, float meanError) const {   // :141-179
    int suspiciousReads[2] = {0, 0};
    size_t adjust = lastReadBiased ? 1 : 0;
    int numberOfReads = 0;
    for (size_t read = 0; read < mostFrequentBases.size() - adjust; read++) {
        int readNumber = moreFrequence[read] + lessFrequence[read];
        float threshold = (float)(0.5 * readNumber + 3 * std::sqrt(readNumber * 0.5 * (1 - 0.5)));
        threshold = std::min(threshold, float(readNumber) - 1);
        if ((float)moreFrequence[read] > threshold) {
            if (mostFrequentBases[read] == -1) { suspiciousReads[0]++; numberOfReads++; }
            else if (mostFrequentBases[read] == 1) { suspiciousReads[1]++; numberOfReads++; }
        }
    }
    float minNumberOfReads = meanError * numberOfReads / 2;
    if (suspiciousReads[0] < minNumberOfReads || suspiciousReads[1] < minNumberOfReads) return false;
    return true;
}

static double comb_lgamma_log(double n, double k) {                           // :186-188
    return std::lgamma(n + 1) - std::lgamma(k + 1) - std::lgamma(n - k + 1);
}

float Partition::isSignificant(int total_number_of_columns_in_pileup) const {  // :197-233
    int numberOfMutatedReads = 0, numberOfReads = 0, number_of_columns = 0;
    for (int p = 0; (size_t)p < mostFrequentBases.size(); p++) {
        if (mostFrequentBases[p] == -1 && moreFrequence[p] > 1 && lessFrequence[p] == 0) {
            numberOfMutatedReads++;
            if (moreFrequence[p] > number_of_columns) number_of_columns = moreFrequence[p];
        }
        if (p != 0 && p != -2 && moreFrequence[p] > 1 && lessFrequence[p] == 0) numberOfReads++;   // :209 (index, not value)
    }
    double p_value = std::exp(::log((double)(float(numberOfMutatedReads) / numberOfReads)) * number_of_columns * numberOfMutatedReads
                              + comb_lgamma_log(numberOfReads, numberOfMutatedReads)
                              + comb_lgamma_log(total_number_of_columns_in_pileup, number_of_columns));
    return (float)std::max(0.0, p_value);
}

void Partition::augmentPartition(const Column& supp, int pos) {               // :243-397
    if (pos != -1) {
        if (pos < pos_left || pos_left == -1) pos_left = pos;
        if (pos > pos_right) pos_right = pos;
    }
    if (supp.readIdxs.size() == 0) return;

    std::vector<int> content(256, 0);
    for (unsigned char c : supp.content) content[c]++;
    unsigned char mostFrequent = 'b', secondFrequent = 'b';
    int maxFrequence = -1;
    for (int i = 0; i < 255; i++) {
        if (i != ' ' && content[i] > maxFrequence) { mostFrequent = (unsigned char)i; maxFrequence = content[i]; }
    }
    int maxFrequence2 = -1;
    for (int i = 0; i < 255; i++) {
        if (i != mostFrequent && i != ' ') {
            if (content[i] > maxFrequence2) { secondFrequent = (unsigned char)i; maxFrequence2 = content[i]; }
        }
    }
    // phase vote :284-314
    int n1 = 0; size_t n2 = 0; int swapped = 0;
    for (int read : readIdx) {
        while (n2 < supp.readIdxs.size() && supp.readIdxs[n2] < (unsigned)read) n2++;
        if (n2 >= supp.readIdxs.size()) break;
        if (supp.readIdxs[n2] == (unsigned)read) {
            if (supp.content[n2] == mostFrequent && mostFrequentBases[n1] == 1) swapped += 1;
            else if (supp.content[n2] == mostFrequent && mostFrequentBases[n1] == -1) swapped -= 1;
            else if (supp.content[n2] == secondFrequent && mostFrequentBases[n1] == -1) swapped += 1;
            else if (supp.content[n2] == secondFrequent && mostFrequentBases[n1] == 1) swapped -= 1;
        }
        n1++;
    }
    if (swapped < 0) std::swap(mostFrequent, secondFrequent);

    // sorted merge :316-390
    size_t it1 = 0;
    std::vector<int> idxs1_2, more_2, less_2;
    std::vector<short> bases_2;
    n1 = 0; n2 = 0;
    for (unsigned int read : supp.readIdxs) {
        while (it1 != readIdx.size() && (unsigned)readIdx[it1] < read) {
            bases_2.push_back(mostFrequentBases[n1]); more_2.push_back(moreFrequence[n1]); less_2.push_back(lessFrequence[n1]);
            idxs1_2.push_back(readIdx[it1]);
            it1++; n1++;
        }
        short s = 0;
        if (supp.content[n2] == secondFrequent) s = -1;
        if (supp.content[n2] == mostFrequent) s = 1;
        if (it1 == readIdx.size() || (unsigned)readIdx[it1] != read) {       // new read
            n1--;
            bases_2.push_back(s); more_2.push_back(std::abs(s)); less_2.push_back(0);
            idxs1_2.push_back((int)read);
        } else {
            if (mostFrequentBases[n1] == -2 || s == 0) {
                bases_2.push_back(mostFrequentBases[n1]); more_2.push_back(moreFrequence[n1]); less_2.push_back(lessFrequence[n1]);
            } else if (mostFrequentBases[n1] == 0) {
                bases_2.push_back(s); more_2.push_back(1); less_2.push_back(0);
            } else if (s == mostFrequentBases[n1]) {
                bases_2.push_back(mostFrequentBases[n1]); more_2.push_back(moreFrequence[n1] + 1); less_2.push_back(lessFrequence[n1]);
            } else if (s == -mostFrequentBases[n1]) {
                if (lessFrequence[n1] + 1 > moreFrequence[n1]) {
                    bases_2.push_back((short)-mostFrequentBases[n1]); more_2.push_back(moreFrequence[n1] + 1); less_2.push_back(lessFrequence[n1]);
                } else {
                    bases_2.push_back(mostFrequentBases[n1]); more_2.push_back(moreFrequence[n1]); less_2.push_back(lessFrequence[n1] + 1);
                }
            }
            idxs1_2.push_back((int)read);
            it1++;
        }
        n1++; n2++;
    }
    while (it1 != readIdx.size()) {
        bases_2.push_back(mostFrequentBases[n1]); more_2.push_back(moreFrequence[n1]); less_2.push_back(lessFrequence[n1]);
        idxs1_2.push_back(readIdx[it1]);
        it1++; n1++;
    }
    mostFrequentBases = bases_2; moreFrequence = more_2; lessFrequence = less_2; readIdx = idxs1_2;
    numberOfOccurences += 1;
}

void Partition::mergePartition(const Partition& p, short phased) {            // :401-537
    pos_left = std::min(pos_left, p.get_left());
    pos_right = std::max(pos_right, p.get_right());
    const std::vector<int>& moreOther = p.moreFrequence;
    const std::vector<int>& lessOther = p.lessFrequence;
    const std::vector<short>& other = p.mostFrequentBases;
    const std::vector<int>& idx2 = p.readIdx;
    std::vector<int> newIdx, newMore, newLess;
    std::vector<short> newMost;
    size_t n1 = 0, n2 = 0;
    while (n1 < readIdx.size() && n2 < idx2.size()) {
        while (n1 < readIdx.size() && readIdx[n1] < idx2[n2]) {
            newIdx.push_back(readIdx[n1]); newMost.push_back(mostFrequentBases[n1]);
            newMore.push_back(moreFrequence[n1]); newLess.push_back(lessFrequence[n1]); n1++;
        }
        // :430 reads readIdx[n1] even when n1 == size (out of bounds in the reference); whichever way that
        // comparison falls the remaining idx2 entries are appended unchanged, here or in the tail loop.
        while (n2 < idx2.size() && n1 < readIdx.size() && readIdx[n1] > idx2[n2]) {
            newIdx.push_back(idx2[n2]); newMost.push_back((short)(other[n2] * phased));
            newMore.push_back(moreOther[n2]); newLess.push_back(lessOther[n2]); n2++;
        }
        if (n1 < readIdx.size() && n2 < idx2.size() && readIdx[n1] == idx2[n2]) {
            newIdx.push_back(readIdx[n1]);
            if (mostFrequentBases[n1] == 0 || other[n2] == -2) {
                newMost.push_back((short)(other[n2] * phased)); newMore.push_back(moreOther[n2]); newLess.push_back(lessOther[n2]);
            } else if (other[n2] == 0 || mostFrequentBases[n1] == -2) {
                newMost.push_back(mostFrequentBases[n1]); newMore.push_back(moreFrequence[n1]); newLess.push_back(lessFrequence[n1]);
            } else if (phased * other[n2] == mostFrequentBases[n1]) {
                int whichone = 0;
                double confidence1 = double(moreFrequence[n1]) / (moreFrequence[n1] + lessFrequence[n1]);
                double confidence2 = double(moreOther[n2]) / (moreOther[n2] + lessOther[n2]);
                if (confidence1 < 0.9 && confidence2 > 0.9 && moreOther[n2] >= 10) whichone = 1;
                else if (confidence2 < 0.9 && confidence1 > 0.9 && moreFrequence[n1] >= 10) whichone = 2;
                newMost.push_back(mostFrequentBases[n1]); newMore.push_back(0); newLess.push_back(0);
                if (whichone != 1) { newMore.back() += moreFrequence[n1]; newLess.back() += lessFrequence[n1]; }
                if (whichone != 2) { newMore.back() += moreOther[n2]; newLess.back() += lessOther[n2]; }
            } else if (phased * other[n2] == -mostFrequentBases[n1]) {
                int whichone = 0;
                double confidence1 = double(moreFrequence[n1]) / (moreFrequence[n1] + lessFrequence[n1]);
                double confidence2 = double(moreOther[n2]) / (moreOther[n2] + lessOther[n2]);
                if (confidence1 < 0.8 && confidence2 > 0.8 && moreOther[n2] >= 10) whichone = 1;
                else if (confidence2 < 0.8 && confidence1 > 0.8 && moreFrequence[n1] >= 10) whichone = 2;
                newMost.push_back(mostFrequentBases[n1]); newMore.push_back(0); newLess.push_back(0);
                if (whichone != 1) { newMore.back() += moreFrequence[n1]; newLess.back() += lessFrequence[n1]; }
                if (whichone != 2) { newMore.back() += lessOther[n2]; newLess.back() += moreOther[n2]; }
                if (newLess.back() > newMore.back()) {
                    newMost.back() = (short)(newMost.back() * -1);
                    std::swap(newMore.back(), newLess.back());
                }
            }
            // (a value outside the four cases above pushes an index without data in the reference; it cannot
            //  happen here because bases are in {-1,0,1} on this path: no mask is ever applied)
            n1++; n2++;
        }
    }
    while (n2 < idx2.size()) {
        newIdx.push_back(idx2[n2]); newMost.push_back((short)(other[n2] * phased));
        newMore.push_back(moreOther[n2]); newLess.push_back(lessOther[n2]); n2++;
    }
    while (n1 < readIdx.size()) {
        newIdx.push_back(readIdx[n1]); newMost.push_back(mostFrequentBases[n1]);
        newMore.push_back(moreFrequence[n1]); newLess.push_back(lessFrequence[n1]); n1++;
    }
    readIdx = newIdx; mostFrequentBases = newMost; lessFrequence = newLess; moreFrequence = newMore;
    numberOfOccurences += p.number();
}

std::vector<float> Partition::getConfidence() const {                        // :811-827
    std::vector<float> conf;
    for (size_t i = 0; i < moreFrequence.size(); i++) {
        if (mostFrequentBases[i] == 0) conf.push_back(0.5);
        else if (moreFrequence[i] + lessFrequence[i] > 0) conf.push_back(float(moreFrequence[i]) / (moreFrequence[i] + lessFrequence[i]));
        else conf.push_back(1);
    }
    return conf;
}

float Partition::compute_conf() {                                            // :716-732
    double conf = 1;
    int numberReads = 0;
    std::vector<float> confidences = getConfidence();
    for (size_t c = 0; c < confidences.size(); c++) {
        if (moreFrequence[c] > 1) { conf *= confidences[c]; numberReads++; }
    }
    if (conf == 1) conf = 0.99;
    double x = 1 / (1 - std::exp(std::log(conf) / numberReads));
    conf_score = (float)(x * x * this->number());   // pow(x,2): gcc expands it to x*x
    return conf_score;
}

// call_variants.cpp:778-967
DistRes distance(const Partition& par1, const Column& par2, char ref_base) {
    DistRes res;
    res.augmented = true;
    const std::vector<int>& idxs1 = par1.readIdx;
    const std::vector<short>& part1 = par1.mostFrequentBases;
    const std::vector<int>& more1 = par1.moreFrequence;
    const std::vector<int>& less1 = par1.lessFrequence;
    const std::vector<unsigned int>& idxs2 = par2.readIdxs;
    const std::vector<unsigned char>& part2 = par2.content;

    float numberOfBases = 0;
    RHMap<unsigned char, int> content2;
    size_t n2 = 0, n1 = 0;
    for (size_t r = 0; r < idxs2.size(); r++) {
        while (n1 < idxs1.size() && (unsigned)idxs1[n1] < idxs2[n2]) n1++;
        if (n1 >= idxs1.size()) break;
        if ((unsigned)idxs1[n1] == idxs2[n2] && part1[n1] != -2) {
            if (!content2.contains(part2[n2])) content2[part2[n2]] = 0;
            numberOfBases += 1;
            content2[part2[n2]] += 1;
        }
        n2++;
    }
    if (numberOfBases == 0) { res.augmented = false; return res; }          // :817-828

    unsigned char mostFrequent = (unsigned char)ref_base;
    (void)content2[(unsigned char)ref_base];                                 // :833 inserts the key if absent
    unsigned char secondFrequent = ' ';
    int maxFrequence2 = -1;
    content2.for_each([&](unsigned char k, int v) {
        if ((int)ref_base != (int)k) {                                        // :838 signed char vs unsigned char
            if (v > maxFrequence2) { secondFrequent = k; maxFrequence2 = v; }
        }
    });

    Column& np = res.partition_to_augment;
    np.readIdxs = par2.readIdxs;
    np.content.resize(par2.content.size());
    for (size_t ci = 0; ci < par2.content.size(); ci++) {
        unsigned char c = par2.content[ci];
        np.content[ci] = c == mostFrequent ? 'A' : c == secondFrequent ? 'a' : ' ';
    }

    int m00 = 0, m01 = 0, m10 = 0, m11 = 0, s11 = 0, s10 = 0, s01 = 0, s00 = 0;
    size_t i1 = 0, i2 = 0;
    while (i1 < idxs1.size() && i2 < idxs2.size()) {
        if ((unsigned)idxs1[i1] == idxs2[i2]) {
            bool solid = less1[i1] <= 1 && more1[i1] >= 3;
            if (part2[i2] == mostFrequent) {
                if (part1[i1] == 1) { m11++; if (solid) s11++; }
                else if (part1[i1] == -1) { m01++; if (solid) s01++; }
            } else if (part2[i2] == secondFrequent) {
                if (part1[i1] == 1) { m10++; if (solid) s10++; }
                else if (part1[i1] == -1) { m00++; if (solid) s00++; }
            }
            i1++; i2++;
        } else if (idxs2[i2] > (unsigned)idxs1[i1]) i1++;
        else i2++;
    }
    res.n00 = m00; res.n01 = m01; res.n10 = m10; res.n11 = m11;
    res.solid10 = s10; res.solid11 = s11; res.solid00 = s00; res.solid01 = s01;
    res.phased = 1;
    res.secondBase = secondFrequent;
    return res;
}

// call_variants.cpp:977-1127
DistRes distance(const Partition& par1, const Partition& par2, int threshold_p) {
    int numberOfComparableBases = 0;
    const std::vector<int>& idx1 = par1.readIdx; const std::vector<int>& idx2 = par2.readIdx;
    const std::vector<short>& part1 = par1.mostFrequentBases; const std::vector<short>& part2 = par2.mostFrequentBases;
    const std::vector<int>& more1 = par1.moreFrequence; const std::vector<int>& less1 = par1.lessFrequence;
    const std::vector<int>& more2 = par2.moreFrequence; const std::vector<int>& less2 = par2.lessFrequence;
    int scores[2] = {0, 0};
    short ndivergentPositions[2] = {0, 0};
    short nNotSurePositions[2] = {0, 0};
    int matches00[2] = {0, 0}, matches01[2] = {0, 0}, matches10[2] = {0, 0}, matches11[2] = {0, 0};
    size_t r1 = 0, r2 = 0;
    while (r1 < idx1.size() && r2 < idx2.size()) {
        if (idx1[r1] < idx2[r2]) r1++;
        else if (idx2[r2] < idx1[r1]) r2++;
        else if (more1[r1] > 1 && more2[r2] > 1) {
            numberOfComparableBases += 1;
            float threshold1 = (float)(0.5 * (more1[r1] + less1[r1]) + 3 * std::sqrt((more1[r1] + less1[r1]) * 0.5 * (1 - 0.5)));
            float threshold2 = (float)(0.5 * (more2[r2] + less2[r2]) + 3 * std::sqrt((more2[r2] + less2[r2]) * 0.5 * (1 - 0.5)));
            bool both = (float)more1[r1] > threshold1 && (float)more2[r2] > threshold2;
            bool either = (float)more1[r1] > threshold1 || (float)more2[r2] > threshold2;
            if (part2[r2] == 1) {
                if (part1[r1] == 1) {
                    scores[0] += 1; scores[1] -= 1; matches11[0] += 1; matches10[1] += 1;
                    if (both) ndivergentPositions[1] += 1;
                    if (either) nNotSurePositions[1] += 1;
                } else if (part1[r1] == -1) {
                    scores[0] -= 1; scores[1] += 1; matches01[0] += 1; matches00[1] += 1;
                    if (both) ndivergentPositions[0] += 1;
                    if (either) nNotSurePositions[0] += 1;
                }
            } else if (part2[r2] == -1) {
                if (part1[r1] == 1) {
                    scores[0] -= 1; scores[1] += 1; matches10[0] += 1; matches11[1] += 1;
                    if (both) ndivergentPositions[0] += 1;
                    if (either) nNotSurePositions[0] += 1;
                } else if (part1[r1] == -1) {
                    scores[0] += 1; scores[1] -= 1; matches00[0] += 1; matches01[1] += 1;
                    if (both) ndivergentPositions[1] += 1;
                    if (either) nNotSurePositions[1] += 1;
                }
            }
            r1++; r2++;
        } else { r1++; r2++; }
    }
    DistRes res;
    res.augmented = true;
    if ((ndivergentPositions[0] >= threshold_p && ndivergentPositions[1] >= threshold_p)
        || (nNotSurePositions[0] >= 5 && nNotSurePositions[1] >= 5) || numberOfComparableBases == 0)
        res.augmented = false;
    int maxScoreIdx = 0;
    if (scores[1] > scores[0]) maxScoreIdx = 1;
    res.n00 = matches00[maxScoreIdx]; res.n01 = matches01[maxScoreIdx];
    res.n10 = matches10[maxScoreIdx]; res.n11 = matches11[maxScoreIdx];
    res.phased = (short)(-2 * maxScoreIdx + 1);
    return res;
}

// call_variants.cpp:1135-1163
float computeChiSquare(const DistRes& dis) {
    int n = dis.n00 + dis.n01 + dis.n10 + dis.n11;
    if (n == 0) return 0;
    float pmax1 = float(dis.n10 + dis.n11) / n;
    float pmax2 = float(dis.n01 + dis.n11) / n;
    if (pmax1 * (1 - pmax1) == 0 && pmax2 * (1 - pmax2) == 0) return -1;
    if (pmax1 * pmax2 * (1 - pmax1) * (1 - pmax2) == 0) return 0;
    // float marginal products; pow(float,int) promotes to double and gcc expands ^2 to a multiply
    float e00 = (1 - pmax1) * (1 - pmax2) * n, e01 = (1 - pmax1) * pmax2 * n;
    float e10 = pmax1 * (1 - pmax2) * n, e11 = pmax1 * pmax2 * n;
    double d00 = (double)(float)(dis.n00 - e00), d01 = (double)(float)(dis.n01 - e01);
    double d10 = (double)(float)(dis.n10 - e10), d11 = (double)(float)(dis.n11 - e11);
    float res = (float)(d00 * d00 / (double)e00 + d01 * d01 / (double)e01 + d10 * d10 / (double)e10 + d11 * d11 / (double)e11);
    return res;
}

// call_variants.cpp:577-768
void keep_only_robust_variants(std::vector<Column>& msa, std::vector<Column>& snps_in,
                               std::vector<Column>& snps_out, float mean_error,
                               std::vector<Partition>& parts) {
    snps_out.clear();
    std::vector<Partition> partitions;
    int lastposition = -5;
    for (const Column& snp : snps_in) {                                     // loop A :590-638
        if (snp.pos - lastposition <= 5) continue;
        bool found = false;
        int position = snp.pos;
        int number_of_correlating_snps = 0;
        for (size_t p = 0; p < partitions.size(); p++) {
            if (std::abs(snp.pos - partitions[p].get_right()) > 50000) continue;
            DistRes dis = distance(partitions[p], snp, (char)snp.ref_base);
            int comparable = dis.n00 + dis.n11 + dis.n01 + dis.n10;
            if (dis.n00 + dis.n01 > 0.1 * comparable && dis.n00 + dis.n01 < 0.9 * comparable
                && dis.n01 + dis.n11 > 0.1 * comparable && dis.n01 + dis.n11 < 0.9 * comparable
                && computeChiSquare(dis) > 15) {
                number_of_correlating_snps += 1;
                partitions[p].number_of_correlating_snps += 1;
            }
            if ((dis.n01 <= std::max(0.1 * (dis.n00 + dis.n01), 1.0) && dis.n10 < std::max(0.1 * (dis.n11 + dis.n10), 1.0) && (size_t)comparable >= snp.readIdxs.size() / 2)
                || (dis.n00 <= std::max(0.1 * (dis.n00 + dis.n01), 1.0) && dis.n11 < std::max(0.1 * (dis.n11 + dis.n10), 1.0) && (size_t)comparable >= snp.readIdxs.size() / 2)) {
                found = true;
                partitions[p].augmentPartition(dis.partition_to_augment, position);
                break;
            }
        }
        if (!found) {
            Partition p(snp, position, snp.ref_base);
            p.number_of_correlating_snps = number_of_correlating_snps;
            partitions.push_back(p);
        } else {
            lastposition = snp.pos;
        }
    }
    if (partitions.size() == 0) return;

    std::vector<Partition> finals;                                          // loop B :646-708
    for (size_t p1 = 0; p1 < partitions.size(); p1++) {
        double p_value = partitions[p1].isSignificant((int)snps_in.size());
        if ((p_value < 0.001 || partitions[p1].number_of_correlating_snps > 1) && partitions[p1].isInformative(false, mean_error)) {
            bool different = true;
            for (size_t p2 = 0; p2 < finals.size(); p2++) {
                DistRes dis = distance(finals[p2], partitions[p1], 2);
                if (dis.augmented
                    && (dis.n00 + dis.n11 > 5 * (dis.n01 + dis.n10) || dis.n10 + dis.n01 > 5 * (dis.n00 + dis.n11))
                    && dis.n10 < std::max(2, 2 * dis.n01) && dis.n01 < std::max(2, 2 * dis.n10)) {
                    Partition newPart = finals[p2];
                    newPart.mergePartition(partitions[p1], dis.phased);
                    if (dis.n01 + dis.n10 < 0.1 * (dis.n00 + dis.n11) || newPart.compute_conf() > finals[p2].compute_conf()) {
                        finals[p2].mergePartition(partitions[p1], dis.phased);
                        different = false;
                        break;
                    }
                }
            }
            if (different) finals.push_back(partitions[p1]);
        }
    }

    for (const Column& snp : snps_in) {                                     // loop C :721-738
        for (size_t p = 0; p < finals.size(); p++) {
            DistRes dis = distance(finals[p], snp, (char)snp.ref_base);
            float chisqu = computeChiSquare(dis);
            if (dis.n00 + dis.n01 + dis.n10 + dis.n11 > 0.5 * snp.content.size() && chisqu > 15) {
                snps_out.push_back(snp);
                break;
            }
        }
    }

    size_t idxSnps = 0;                                                     // loop D :741-764
    std::vector<Column> snps_out_tmp = snps_out;
    snps_out.clear();
    for (int position = 0; (size_t)position < msa.size(); position++) {
        if (idxSnps < snps_out_tmp.size() && snps_out_tmp[idxSnps].pos == position) {
            snps_out.push_back(snps_out_tmp[idxSnps]);
            idxSnps++;
        } else {
            int rb = msa[position].ref_base, sb = msa[position].second_base;
            if (rb % 5 != sb % 5 && ((sb - '!') % 5 != 4 || (sb / 5 % 5 != rb % 5 && sb / 25 % 5 != rb % 5))) {
                for (size_t p = 0; p < finals.size(); p++) {
                    DistRes dis = distance(finals[p], msa[position], (char)msa[position].ref_base);
                    if (computeChiSquare(dis) > 20.0 && dis.n10 + dis.n00 > 4 && dis.n01 + dis.n11 > 4) {
                        snps_out.push_back(msa[position]);
                        break;
                    }
                }
            }
        }
    }
    parts = finals;
}

// call_variants.cpp:1300-1354
ContigVariants call_variants_on_contig(const Contig& c, const std::vector<std::string>& read_seq,
                                       float automatic_snp_threshold) {
    ContigVariants out;
    MsaResult msa = generate_msa(c, read_seq);
    out.meanDistance = msa.meanDistance;
    float meanDistance = msa.meanDistance;
    CallResult cr = call_variants(msa.cols, msa.newref, meanDistance, automatic_snp_threshold);
    out.depth = cr.depth;
    for (auto& s : cr.suspicious) out.candidate_pos.push_back(s.pos);
    for (auto& s : cr.automatic) out.automatic_pos.push_back(s.pos);
    std::vector<Column> filtered;
    keep_only_robust_variants(msa.cols, cr.suspicious, filtered, meanDistance, out.partitions);
    for (auto& s : filtered) out.filtered_pos.push_back(s.pos);
    // two-pointer merge that stops when either list ends :1335-1352
    size_t ia = 0, ifi = 0;
    while (ia < cr.automatic.size() && ifi < filtered.size()) {
        if (cr.automatic[ia].pos < filtered[ifi].pos) { out.merged.push_back(cr.automatic[ia]); ia++; }
        else if (cr.automatic[ia].pos > filtered[ifi].pos) { out.merged.push_back(filtered[ifi]); ifi++; }
        else { out.merged.push_back(cr.automatic[ia]); ia++; ifi++; }
    }
    return out;
}

}  // namespace hso
