/* orc_profile.c -- TEST INFRASTRUCTURE ONLY (never linked into libparasuite_hip.so): CPU restatement, line by line, of
 * utils.errorprofile.ErrorProfiling.inferErrorProfile -- /root/reference/src/src/utils/errorprofile/ErrorProfiling.java:100-631.
 * The Java source IS in the reference tree, so every block below cites the lines it follows; the restatement is still
 * "parity unpinned": the reference holds no fixture (input BAM + expected .errorprofile) and no JVM exists here to make one.
 * Input: SAM text (the Java reads BAM through htsjdk: same fields) and the FASTA itself (the Java reads it through
 * IndexedFastaSequenceFile), so nothing of the product's index or BAM code is shared with the thing under test.
 * Only the two files the mapping step consumes are written: <out>.errorprofile (:504-531) and <out>.indelprofile (:545-591). */
#include <ctype.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { char *name; char *seq; long len; } contig_t;

static char g_err[4400];
const char *orc_profile_last_error(void) { return g_err; }

/* ErrorProfiling.java:634-664 calculateArrayPos */
static int array_pos(unsigned char b)
{
    switch (b) { case 65: case 97: return 0; case 67: case 99: return 1; case 71: case 103: return 2; case 84: case 116: return 3; default: return -1; }
}
/* htsjdk SequenceUtil.reverseComplement(byte[]): reversed, a<->t c<->g (case kept), every other byte as it is */
static unsigned char compl(unsigned char b)
{
    switch (b) { case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
                 case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return b; }
}
static void revcomp(unsigned char *s, int n)
{
    int i;
    for (i = 0; i < n / 2; ++i) { unsigned char a = compl(s[i]), b = compl(s[n - 1 - i]); s[i] = b; s[n - 1 - i] = a; }
    if (n & 1) s[n / 2] = compl(s[n / 2]);
}

/* java.lang.Double.toString (JDK 19+: shortest digits that read back) */
static void java_double(double v, char *out)
{
    char buf[64], dig[32]; int prec, e10, nd = 0, neg = 0; char *ep; size_t i;
    if (v != v) { strcpy(out, "NaN"); return; }
    if (isinf(v)) { strcpy(out, v > 0 ? "Infinity" : "-Infinity"); return; }
    if (v == 0) { strcpy(out, signbit(v) ? "-0.0" : "0.0"); return; }
    for (prec = 1; prec <= 17; ++prec) { snprintf(buf, sizeof buf, "%.*e", prec - 1, v); if (strtod(buf, NULL) == v) break; }
    ep = strchr(buf, 'e'); e10 = atoi(ep + 1);
    for (i = 0; buf + i < ep; ++i) { if (buf[i] == '-') neg = 1; else if (isdigit((unsigned char)buf[i])) dig[nd++] = buf[i]; }
    while (nd > 1 && dig[nd - 1] == '0') --nd;
    dig[nd] = 0;
    out[0] = 0;
    if (neg) strcat(out, "-");
    if (fabs(v) >= 1e-3 && fabs(v) < 1e7) {
        if (e10 >= 0) {
            int k; char *o = out + strlen(out);
            for (k = 0; k <= e10; ++k) *o++ = k < nd ? dig[k] : '0';
            *o++ = '.';
            if (nd > e10 + 1) { for (k = e10 + 1; k < nd; ++k) *o++ = dig[k]; } else *o++ = '0';
            *o = 0;
        } else {
            int k; char *o = out + strlen(out);
            *o++ = '0'; *o++ = '.';
            for (k = 0; k < -e10 - 1; ++k) *o++ = '0';
            for (k = 0; k < nd; ++k) *o++ = dig[k];
            *o = 0;
        }
    } else {
        char *o = out + strlen(out); int k;
        *o++ = dig[0]; *o++ = '.';
        if (nd > 1) { for (k = 1; k < nd; ++k) *o++ = dig[k]; } else *o++ = '0';
        sprintf(o, "E%d", e10);
    }
}
void orc_java_double(double v, char *out) { java_double(v, out); }

static contig_t *load_fasta(const char *path, int *n_out)
{
    FILE *f = fopen(path, "rb"); contig_t *c = NULL; int n = 0, cap = 0; char *line = NULL; size_t lcap = 0; long m;
    long scap = 0;
    if (!f) return NULL;
    while ((m = getline(&line, &lcap, f)) >= 0) {
        while (m > 0 && (line[m - 1] == '\n' || line[m - 1] == '\r')) line[--m] = 0;
        if (line[0] == '>') {
            char *e = line + 1;
            if (n == cap) { cap = cap ? 2 * cap : 8; c = (contig_t *)realloc(c, sizeof(contig_t) * (size_t)cap); }
            while (*e && !isspace((unsigned char)*e)) ++e;
            *e = 0;
            c[n].name = strdup(line + 1); c[n].seq = NULL; c[n].len = 0; scap = 0; ++n;
        } else if (n) {
            contig_t *q = &c[n - 1];
            if (q->len + m + 1 > scap) { scap = 2 * (q->len + m + 1); q->seq = (char *)realloc(q->seq, (size_t)scap); }
            memcpy(q->seq + q->len, line, (size_t)m); q->len += m;
        }
    }
    free(line); fclose(f);
    *n_out = n;
    return c;
}

/* returns the number of records processed (numReadsProcessed), -1 on error */
long orc_error_profile(const char *sam_path, const char *fasta_path, int maxReadLength, const char *out_prefix)
{
    int n_contigs = 0, i, j, k;
    contig_t *ctg = load_fasta(fasta_path, &n_contigs);
    FILE *f = fopen(sam_path, "rb");
    char *line = NULL; size_t lcap = 0; long m;
    /* ErrorProfiling.java:58-87 */
    long *positionConversions = (long *)calloc((size_t)maxReadLength * 16, sizeof(long));
    double *insertionsPerPos = (double *)calloc((size_t)maxReadLength, sizeof(double));
    double *deletionsPerPos = (double *)calloc((size_t)maxReadLength, sizeof(double));
    long *totalCountsPerPos = (long *)calloc((size_t)maxReadLength, sizeof(long));
    double totalErrorCounts[4][4] = {{0}}, totalBaseCounts[4] = {0, 0, 0, 0};
    long numReadsProcessed = 0;
    char path[4096], num[64];
    if (!ctg || !f) { snprintf(g_err, sizeof g_err, "cannot open %s", !ctg ? fasta_path : sam_path); return -1; }
    while ((m = getline(&line, &lcap, f)) >= 0) {                       /* :145 for (SAMRecord readHit : samFileReader) */
        char *fld[11], *p = line; int nf = 0, flag, L, R = 0, cid = -1, mappingLength, skip = 0;
        long start, end;
        unsigned char *readSequence, *refSequenceForRead; int read_len, ref_len;
        if (line[0] == '@') continue;
        while (m > 0 && (line[m - 1] == '\n' || line[m - 1] == '\r')) line[--m] = 0;
        while (nf < 11 && p) { fld[nf++] = p; p = strchr(p, '\t'); if (p) *p++ = 0; }
        if (nf < 11) continue;
        flag = atoi(fld[1]); start = atol(fld[3]);
        if (flag & 4) continue;                                           /* :155-158 unmapped */
        if (flag & 1024) continue;                                        /* :159-162 duplicate */
        if (start == 0) continue;                                         /* :163-166 */
        L = (int)strlen(fld[9]);
        { const char *c = fld[5]; while (*c) { long len = strtol(c, (char **)&c, 10); char op = *c++; if (op == 'M' || op == 'D' || op == 'N' || op == '=' || op == 'X') R += (int)len; } }
        end = start + R - 1;                                              /* htsjdk getAlignmentEnd */
        for (i = 0; i < n_contigs; ++i) if (strcmp(ctg[i].name, fld[2]) == 0) { cid = i; break; }
        if (cid < 0 || start < 1 || end > ctg[cid].len) { snprintf(g_err, sizeof g_err, "record outside the reference: %s:%ld", fld[2], start); return -1; }
        read_len = L; ref_len = R;
        readSequence = (unsigned char *)malloc((size_t)(L > R ? L : R) + 8);
        refSequenceForRead = (unsigned char *)malloc((size_t)(L > R ? L : R) + 8);
        memcpy(readSequence, fld[9], (size_t)L);                          /* :168 getReadBases */
        memcpy(refSequenceForRead, ctg[cid].seq + start - 1, (size_t)R);  /* :169-172 getSubsequenceAt(start, end) */
        ++numReadsProcessed;                                              /* :174 */
        if (refSequenceForRead[0] == 0) { free(readSequence); free(refSequenceForRead); continue; }   /* :180 */
        mappingLength = read_len > ref_len ? read_len : ref_len;          /* :188-193 */
        if (read_len != ref_len) {                                        /* :194 */
            unsigned char *refT = (unsigned char *)calloc((size_t)mappingLength + 8, 1), *readT = (unsigned char *)calloc((size_t)mappingLength + 8, 1);
            int passedRef = 0, passedRead = 0, passedMatches = 0, z, q;
            const char *c = fld[5];
            while (*c) {                                                  /* :206 for (CigarElement elem ...) */
                int len = (int)strtol(c, (char **)&c, 10); char op = *c++;
                if (op == 'M' || op == 'X' || op == '=') {                /* :214-243 */
                    for (z = 0; z < len; ++z) {
                        if (z + passedMatches >= mappingLength || z + passedRef >= ref_len || z + passedRead >= read_len) { skip = 1; continue; }   /* the caught ArrayIndexOutOfBoundsException */
                        refT[z + passedMatches] = refSequenceForRead[z + passedRef];
                        readT[z + passedMatches] = readSequence[z + passedRead];
                    }
                    passedMatches += len; passedRef += len; passedRead += len;
                } else if (op == 'N') { passedRef += len; passedRead += len; }          /* :244-248 */
                else if (op == 'I') {                                     /* :249-268 */
                    for (z = 0; z < len; ++z) if (passedMatches + z < mappingLength) refT[passedMatches + z] = 45;
                    passedMatches += len; passedRead += len;
                    for (q = 1; q <= len; ++q) if (passedMatches + q < maxReadLength) insertionsPerPos[passedMatches + q] += 1.0;
                } else if (op == 'D') {                                   /* :270-288 */
                    for (z = 0; z < len; ++z) if (passedMatches + z < mappingLength) readT[passedMatches + z] = 45;
                    passedMatches += len; passedRef += len;
                    for (q = 1; q <= len; ++q) if (passedMatches + q < maxReadLength) deletionsPerPos[passedMatches + q] += 1.0;
                }
            }
            free(refSequenceForRead); free(readSequence);
            refSequenceForRead = refT; readSequence = readT;              /* :292-293 */
            read_len = ref_len = mappingLength;
        }
        if (skip) { free(readSequence); free(refSequenceForRead); continue; }           /* :298-301 */
        if (flag & 16) { revcomp(readSequence, read_len); revcomp(refSequenceForRead, ref_len); }   /* :306-311 */
        if (read_len > maxReadLength) { snprintf(g_err, sizeof g_err, "read longer than maxReadLength"); return -1; }   /* the Java's array bound */
        for (i = 0; i < read_len; ++i) {                                  /* :342-409 */
            int arrayPosRef = array_pos(refSequenceForRead[i]), arrayPosRead = array_pos(readSequence[i]);
            if (arrayPosRef >= 0 && arrayPosRead >= 0) ++positionConversions[(size_t)i * 16 + arrayPosRef * 4 + arrayPosRead];   /* :376-377 */
        }
        free(readSequence); free(refSequenceForRead);
    }
    free(line); fclose(f);
    /* :448-458 */
    for (i = 0; i < maxReadLength; ++i)
        for (j = 0; j < 4; ++j)
            for (k = 0; k < 4; ++k) {
                long x = positionConversions[(size_t)i * 16 + j * 4 + k];
                totalErrorCounts[j][k] += (double)x; totalBaseCounts[j] += (double)x; totalCountsPerPos[i] += x;
            }
    /* :504-531 */
    snprintf(path, sizeof path, "%s.errorprofile", out_prefix);
    f = fopen(path, "wb");
    if (!f) { snprintf(g_err, sizeof g_err, "cannot write %s", path); return -1; }
    for (j = 0; j < 4; ++j) {
        for (k = 0; k < 4; ++k) { java_double(totalErrorCounts[j][k] / totalBaseCounts[j], num); fprintf(f, "%s\t", num); }
        fputc('\n', f);
    }
    fclose(f);
    /* :545-591 */
    {
        double insertionsOverall = 0.0, deletionsOverall = 0.0; int insertionsZero = 0, deletionsZero = 0;
        for (i = 0; i < maxReadLength; ++i) {
            if (totalCountsPerPos[i] == 0) { insertionsPerPos[i] = 0.0; deletionsPerPos[i] = 0.0; ++insertionsZero; ++deletionsZero; }
            else {
                insertionsPerPos[i] = insertionsPerPos[i] / (double)totalCountsPerPos[i];
                if (insertionsPerPos[i] > 0) insertionsOverall += insertionsPerPos[i]; else ++insertionsZero;
                deletionsPerPos[i] = deletionsPerPos[i] / (double)totalCountsPerPos[i];
                if (deletionsPerPos[i] > 0) deletionsOverall += deletionsPerPos[i]; else ++deletionsZero;
            }
        }
        if (maxReadLength == insertionsZero && maxReadLength == deletionsZero) { insertionsOverall = 0.0; deletionsOverall = 0.0; }
        else { insertionsOverall = insertionsOverall / (double)(maxReadLength - insertionsZero); deletionsOverall = deletionsOverall / (double)(maxReadLength - deletionsZero); }
        snprintf(path, sizeof path, "%s.indelprofile", out_prefix);
        f = fopen(path, "wb");
        if (!f) { snprintf(g_err, sizeof g_err, "cannot write %s", path); return -1; }
        java_double(insertionsOverall, num); fprintf(f, "%s\t", num);
        java_double(deletionsOverall, num); fprintf(f, "%s", num);
        fclose(f);
    }
    for (i = 0; i < n_contigs; ++i) { free(ctg[i].name); free(ctg[i].seq); }
    free(ctg); free(positionConversions); free(insertionsPerPos); free(deletionsPerPos); free(totalCountsPerPos);
    return numReadsProcessed;
}
