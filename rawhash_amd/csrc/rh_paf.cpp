// PAF line formatting (host): the 12 columns + tags that step 2 of the reference pipeline prints
// (rmap.cpp:740-783; tag strings assembled at rmap.cpp:523-571).  Tag order mt, ci, sl, cm, nc, s1, sm is part of the
// de-facto format (test/scripts/compare_pafs.py parses by column index).  `sm` is always 0.00 in the reference.
#include "rh_index.h"

extern "C" int rh_paf_format(const rh_index *ix, const rh_map_record_t *r, const char *read_name, double mt_ms, char *buf, size_t cap)
{
	std::string tags;
	char t[160];
	snprintf(t, sizeof(t), "mt:f:%.6f\tci:i:%d\tsl:i:%d", mt_ms, r->tag_ci, r->tag_sl);
	tags = t;
	if (r->mapped || r->tag_nc >= 1) {
		snprintf(t, sizeof(t), "\tcm:i:%d\tnc:i:%d\ts1:i:%d\tsm:f:%.2f", r->tag_cm, r->tag_nc, r->tag_s1, 0.0);
		tags += t;
	} else tags += "\tcm:i:0\tnc:i:0\ts1:i:0\tsm:f:0";   // no chain at all: literal zeros, note "sm:f:0" (rmap.cpp:539-544)
	int n;
	if (r->mapped) {
		if (r->ref_id >= ix->names.size()) { if (cap) buf[0] = 0; return 0; }   // silently not printed (rmap.cpp:750)
		n = snprintf(buf, cap, "%s\t%u\t%u\t%u\t%c\t%s\t%u\t%u\t%u\t%u\t%u\t%u\t%s\n", read_name,
		             r->read_length, r->read_start_position, r->read_end_position, r->rev ? '-' : '+',
		             ix->names[r->ref_id].c_str(), ix->lens[r->ref_id],
		             r->fragment_start_position, r->fragment_start_position + r->fragment_length,
		             r->read_end_position - r->read_start_position - 1, r->fragment_length, (unsigned)r->mapq, tags.c_str());
	} else {
		n = snprintf(buf, cap, "%s\t%u\t*\t*\t*\t*\t*\t*\t*\t*\t*\t%u\t%s\n", read_name, r->read_length, (unsigned)r->mapq, tags.c_str());
	}
	if (n < 0 || (size_t)n >= cap) { rh_set_error("PAF buffer too small"); return -1; }
	return n;
}
