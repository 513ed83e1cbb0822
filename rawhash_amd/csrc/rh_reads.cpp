// "RHR1" read container (own format; stands in for SLOW5/BLOW5 until a native reader lands).
// The loader produces the SoA/CSR batch the C ABI takes: raw int16 samples + per-read calibration
// {offset (double), scale = (float)(range/digitisation)} exactly as ri_read_sig_slow5 derives them (rsig.c:494).
#include "rh_common.h"
#include <exception>

struct rh_reads_s {
	std::vector<std::string> names;
	std::vector<int16_t> samples;
	std::vector<uint64_t> offsets;
	std::vector<double> cal_offset;
	std::vector<float> cal_scale;
};

static long file_remaining(FILE *fp)
{
	const long at = ftell(fp);
	if (at < 0 || fseek(fp, 0, SEEK_END) != 0) return -1;
	const long end = ftell(fp);
	fseek(fp, at, SEEK_SET);
	return end < at ? -1 : end - at;
}

static rh_reads *reads_load_rhr(const char *path);

extern "C" rh_reads *rh_reads_load(const char *path)
{
	try { return reads_load_rhr(path); }
	catch (const std::exception &e) { rh_set_error("%s: %s", path, e.what()); return 0; }   // (bad_alloc must not cross the C ABI)
}

static rh_reads *reads_load_rhr(const char *path)
{
	FILE *fp = fopen(path, "rb");
	if (!fp) { rh_set_error("cannot open %s", path); return 0; }
	char magic[4]; uint32_t n;
	if (fread(magic, 1, 4, fp) != 4 || memcmp(magic, "RHR1", 4) != 0 || fread(&n, 4, 1, fp) != 1) {
		fclose(fp); rh_set_error("%s: not an RHR1 file", path); return 0;
	}
	rh_reads *r = new rh_reads_s();
	r->offsets.push_back(0);
	for (uint32_t i = 0; i < n; ++i) {
		uint32_t l, ns; double dig, range, off;
		std::string name;
		bool ok = fread(&l, 4, 1, fp) == 1 && (long)l <= file_remaining(fp);      // lengths are checked against what is left of the file
		if (ok) { name.resize(l); ok = l == 0 || fread(&name[0], 1, l, fp) == l; }
		ok = ok && fread(&ns, 4, 1, fp) == 1 && fread(&dig, 8, 1, fp) == 1 && fread(&range, 8, 1, fp) == 1 && fread(&off, 8, 1, fp) == 1;
		ok = ok && (long)ns * 2 <= file_remaining(fp);
		if (ok) {
			size_t o = r->samples.size();
			r->samples.resize(o + ns);
			ok = ns == 0 || fread(&r->samples[o], 2, ns, fp) == ns;
		}
		if (!ok) { fclose(fp); delete r; rh_set_error("%s: truncated at read %u", path, i); return 0; }
		r->names.push_back(name);
		r->offsets.push_back(r->samples.size());
		r->cal_offset.push_back(off);
		r->cal_scale.push_back((float)(range / dig));
	}
	fclose(fp);
	return r;
}

extern "C" void rh_reads_destroy(rh_reads *r) { delete r; }
extern "C" uint32_t rh_reads_n(const rh_reads *r) { return (uint32_t)r->names.size(); }
extern "C" const char *rh_reads_name(const rh_reads *r, uint32_t i) { return i < r->names.size() ? r->names[i].c_str() : 0; }

extern "C" int rh_reads_batch(const rh_reads *r, rh_read_batch_t *out)
{
	memset(out, 0, sizeof(*out));
	out->n_reads = (uint32_t)r->names.size();
	out->samples = r->samples.data();
	out->offsets = r->offsets.data();
	out->cal_offset = r->cal_offset.data();
	out->cal_scale = r->cal_scale.data();
	return 0;
}

extern "C" int rh_reads_write(const char *path, uint32_t n, const char *const *names, const int16_t *samples,
                              const uint64_t *offsets, double digitisation, double range, double offset)
{
	FILE *fp = fopen(path, "wb");
	if (!fp) { rh_set_error("cannot write %s", path); return -1; }
	fwrite("RHR1", 1, 4, fp);
	fwrite(&n, 4, 1, fp);
	for (uint32_t i = 0; i < n; ++i) {
		uint32_t l = (uint32_t)strlen(names[i]), ns = (uint32_t)(offsets[i + 1] - offsets[i]);
		fwrite(&l, 4, 1, fp); fwrite(names[i], 1, l, fp);
		fwrite(&ns, 4, 1, fp);
		fwrite(&digitisation, 8, 1, fp); fwrite(&range, 8, 1, fp); fwrite(&offset, 8, 1, fp);
		fwrite(samples + offsets[i], 2, ns, fp);
	}
	fclose(fp);
	return 0;
}
