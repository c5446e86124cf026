/* oracle/stubs_full/htslib/sam.h -- TEST INFRASTRUCTURE (see hts.h).
 * bam1_t as the SAM/BAM specification (SAMv1 section 4.2) lays a record out: core fields + one data block holding
 * qname, cigar (uint32 op | len << 4), 4-bit packed sequence, qualities, aux.  The accessors below are the ones the
 * reference executes on the call-methylation path (SequenceAlignmentRecord, get_aligned_segments, bam_endpos). */
#ifndef NP_STUBFULL_SAM_H
#define NP_STUBFULL_SAM_H
#include "hts.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct sam_hdr_t { int32_t n_targets; uint32_t* target_len; char** target_name; } sam_hdr_t;
typedef sam_hdr_t bam_hdr_t;
typedef htsFile samFile;
typedef struct bam1_core_t {
    hts_pos_t pos; int32_t tid; uint16_t bin; uint8_t qual; uint8_t l_extranul; uint16_t flag; uint16_t l_qname;
    uint32_t n_cigar; int32_t l_qseq; int32_t mtid; hts_pos_t mpos; hts_pos_t isize;
} bam1_core_t;
typedef struct bam1_t { bam1_core_t core; uint64_t id; uint8_t* data; int l_data; uint32_t m_data; uint32_t mempolicy; } bam1_t;

#define BAM_CMATCH 0
#define BAM_CINS 1
#define BAM_CDEL 2
#define BAM_CREF_SKIP 3
#define BAM_CSOFT_CLIP 4
#define BAM_CHARD_CLIP 5
#define BAM_CPAD 6
#define BAM_CEQUAL 7
#define BAM_CDIFF 8
#define BAM_CBACK 9
#define BAM_CIGAR_STR "MIDNSHP=XB"
#define BAM_CIGAR_SHIFT 4
#define BAM_CIGAR_MASK 0xf
#define BAM_CIGAR_TYPE 0x3C1A7
#define bam_cigar_op(c) ((c) & BAM_CIGAR_MASK)
#define bam_cigar_oplen(c) ((c) >> BAM_CIGAR_SHIFT)
#define bam_cigar_opchr(c) (BAM_CIGAR_STR "??????"[bam_cigar_op(c)])
#define bam_cigar_gen(l, o) ((l) << BAM_CIGAR_SHIFT | (o))
#define bam_cigar_type(o) (BAM_CIGAR_TYPE >> ((o) << 1) & 3)   /* bit 1: consumes query, bit 2: consumes reference */

#define BAM_FPAIRED 1
#define BAM_FPROPER_PAIR 2
#define BAM_FUNMAP 4
#define BAM_FMUNMAP 8
#define BAM_FREVERSE 16
#define BAM_FMREVERSE 32
#define BAM_FREAD1 64
#define BAM_FREAD2 128
#define BAM_FSECONDARY 256
#define BAM_FQCFAIL 512
#define BAM_FDUP 1024
#define BAM_FSUPPLEMENTARY 2048

#define bam_is_rev(b) (((b)->core.flag & BAM_FREVERSE) != 0)
#define bam_is_mrev(b) (((b)->core.flag & BAM_FMREVERSE) != 0)
#define bam_get_qname(b) ((char*)(b)->data)
#define bam_get_cigar(b) ((uint32_t*)((b)->data + (b)->core.l_qname))
#define bam_get_seq(b) ((b)->data + ((b)->core.n_cigar << 2) + (b)->core.l_qname)
#define bam_get_qual(b) ((b)->data + ((b)->core.n_cigar << 2) + (b)->core.l_qname + (((b)->core.l_qseq + 1) >> 1))
#define bam_get_aux(b) ((b)->data + ((b)->core.n_cigar << 2) + (b)->core.l_qname + (((b)->core.l_qseq + 1) >> 1) + (b)->core.l_qseq)
#define bam_get_l_aux(b) ((b)->l_data - ((b)->core.n_cigar << 2) - (b)->core.l_qname - (b)->core.l_qseq - (((b)->core.l_qseq + 1) >> 1))
#define bam_seqi(s, i) ((s)[(i) >> 1] >> ((~(i) & 1) << 2) & 0xf)
#define bam_set_seqi(s, i, b) ((s)[(i) >> 1] = ((s)[(i) >> 1] & (0xf0 >> ((~(i) & 1) << 2))) | ((b) << ((~(i) & 1) << 2)))
extern const char seq_nt16_str[];
extern const unsigned char seq_nt16_table[256];

/* rightmost reference coordinate of the alignment, exclusive: pos + reference-consuming CIGAR lengths (an unmapped or
 * CIGAR-less record spans one base) */
static inline hts_pos_t bam_endpos(const bam1_t* b)
{
    hts_pos_t rlen = 0;
    if (!(b->core.flag & BAM_FUNMAP) && b->core.n_cigar > 0) {
        const uint32_t* cigar = bam_get_cigar(b);
        for (uint32_t k = 0; k < b->core.n_cigar; ++k)
            if (bam_cigar_type(bam_cigar_op(cigar[k])) & 2) rlen += bam_cigar_oplen(cigar[k]);
    }
    if (rlen == 0) rlen = 1;
    return b->core.pos + rlen;
}

bam1_t* bam_init1(void);
void bam_destroy1(bam1_t* b);
bam1_t* bam_dup1(const bam1_t* b);
bam1_t* bam_copy1(bam1_t* bdst, const bam1_t* bsrc);
sam_hdr_t* sam_hdr_read(samFile* fp);
int sam_hdr_write(samFile* fp, const sam_hdr_t* h);
void sam_hdr_destroy(sam_hdr_t* h);
void bam_hdr_destroy(bam_hdr_t* h);
sam_hdr_t* sam_hdr_dup(const sam_hdr_t* h);
sam_hdr_t* bam_hdr_dup(const sam_hdr_t* h);
sam_hdr_t* bam_hdr_init(void);
int sam_hdr_name2tid(sam_hdr_t* h, const char* ref);
int bam_name2id(bam_hdr_t* h, const char* ref);
int sam_read1(samFile* fp, sam_hdr_t* h, bam1_t* b);
int sam_write1(samFile* fp, const sam_hdr_t* h, const bam1_t* b);
hts_idx_t* sam_index_load(htsFile* fp, const char* fn);
hts_idx_t* bam_index_load(const char* fn);
hts_itr_t* sam_itr_queryi(const hts_idx_t* idx, int tid, hts_pos_t beg, hts_pos_t end);
hts_itr_t* sam_itr_querys(const hts_idx_t* idx, sam_hdr_t* hdr, const char* region);
int sam_itr_next(htsFile* htsfp, hts_itr_t* itr, bam1_t* r);
#define sam_open(fn, mode) (hts_open((fn), (mode)))
#define sam_close(fp) hts_close(fp)
#define bam_itr_destroy(iter) hts_itr_destroy(iter)
#define sam_itr_destroy(iter) hts_itr_destroy(iter)
#define bam_itr_queryi(idx, tid, beg, end) sam_itr_queryi(idx, tid, beg, end)
uint8_t* bam_aux_get(const bam1_t* b, const char tag[2]);
int64_t bam_aux2i(const uint8_t* s);
char* bam_aux2Z(const uint8_t* s);
int bam_aux_append(bam1_t* b, const char tag[2], char type, int len, const uint8_t* data);
int bam_aux_update_str(bam1_t* b, const char tag[2], int len, const char* data);
int bam_aux_update_array(bam1_t* b, const char tag[2], uint8_t type, uint32_t items, void* data);
int bam_reg2bin(int64_t beg, int64_t end);
#ifdef __cplusplus
}
#endif
#endif
