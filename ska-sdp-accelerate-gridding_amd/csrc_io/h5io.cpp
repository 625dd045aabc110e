// libgridhip_io.so — native HDF5 I/O for the gridder's datasets.
//
// Exports the same 14 C symbols, with the same signatures and on-disk conventions, as the
// reference's HDF5 shim (/root/reference/hdf5/hdf5.cc:59-186) so that src/Hdf5.hs:30-67 binds it
// unchanged: every call opens and closes the file, names without ".h5" get it appended, complex
// data is a compound {double r; double i;}, dims travel as `int rank, int *dims` in C order.
//
// Written against the core H5F/H5D/H5S/H5T/H5L API.  Departures from the reference, all on the
// error side: the caller's name buffer is never written to (the reference strcat()s ".h5" into
// it), every HDF5 status is checked and the first failure of a call is kept for
// h5io_last_error(); a failed read leaves the output untouched instead of reading through an
// invalid handle.
#include <hdf5.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

struct Cplx {
    double r, i;
};

std::string with_ext(const char *name)
{
    std::string s(name ? name : "");
    if (s.size() < 3 || s.compare(s.size() - 3, 3, ".h5") != 0) s += ".h5";
    return s;
}

void fail(const std::string &what) { if (g_err.empty()) g_err = what; }

struct Quiet {  // silence HDF5's stderr stack dump; errors are reported through h5io_last_error
    H5E_auto2_t fn;
    void *data;
    Quiet()
    {
        H5Eget_auto2(H5E_DEFAULT, &fn, &data);
        H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);
    }
    ~Quiet() { H5Eset_auto2(H5E_DEFAULT, fn, data); }
};

struct File {
    hid_t id;
    File(const char *name, unsigned flags) : id(H5Fopen(with_ext(name).c_str(), flags, H5P_DEFAULT))
    {
        if (id < 0) fail("cannot open " + with_ext(name));
    }
    ~File() { if (id >= 0) H5Fclose(id); }
    bool ok() const { return id >= 0; }
};

enum Kind { K_INT, K_LLONG, K_DOUBLE, K_COMPLEX };

hid_t mem_type(Kind k)  // caller closes the returned handle
{
    switch (k) {
        case K_INT: return H5Tcopy(H5T_NATIVE_INT);
        case K_LLONG: return H5Tcopy(H5T_NATIVE_LLONG);
        case K_DOUBLE: return H5Tcopy(H5T_NATIVE_DOUBLE);
        case K_COMPLEX: {
            hid_t t = H5Tcreate(H5T_COMPOUND, sizeof(Cplx));
            H5Tinsert(t, "r", HOFFSET(Cplx, r), H5T_NATIVE_DOUBLE);
            H5Tinsert(t, "i", HOFFSET(Cplx, i), H5T_NATIVE_DOUBLE);
            return t;
        }
    }
    return -1;
}

size_t elem_size(Kind k) { return k == K_INT ? sizeof(int) : k == K_LLONG ? sizeof(long long) : k == K_DOUBLE ? 8 : 16; }

// shape of a dataset; empty on failure
std::vector<hsize_t> shape_of(hid_t file, const char *dataset)
{
    std::vector<hsize_t> dims;
    hid_t d = H5Dopen2(file, dataset, H5P_DEFAULT);
    if (d < 0) {
        fail(std::string("no dataset ") + dataset);
        return dims;
    }
    hid_t sp = H5Dget_space(d);
    int rank = H5Sget_simple_extent_ndims(sp);
    if (rank >= 0) {
        dims.resize(rank);
        if (rank) H5Sget_simple_extent_dims(sp, dims.data(), nullptr);
    }
    H5Sclose(sp);
    H5Dclose(d);
    return dims;
}

bool read_one(hid_t file, const char *dataset, hid_t mt, void *out)
{
    hid_t d = H5Dopen2(file, dataset, H5P_DEFAULT);
    if (d < 0) {
        fail(std::string("no dataset ") + dataset);
        return false;
    }
    herr_t st = H5Dread(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, out);
    H5Dclose(d);
    if (st < 0) fail(std::string("cannot read ") + dataset);
    return st >= 0;
}

void read_dataset(Kind k, const char *name, const char *dataset, void *out)
{
    g_err.clear();
    Quiet q;
    File f(name, H5F_ACC_RDONLY);
    if (!f.ok() || !out) return;
    hid_t mt = mem_type(k);
    read_one(f.id, dataset, mt, out);
    H5Tclose(mt);
}

// NULL-terminated list of equally shaped datasets, concatenated along a new leading axis
void read_datasets(Kind k, const char *name, char **datasets, void *out)
{
    g_err.clear();
    Quiet q;
    File f(name, H5F_ACC_RDONLY);
    if (!f.ok() || !out || !datasets || !datasets[0]) return;
    std::vector<hsize_t> dims = shape_of(f.id, datasets[0]);
    size_t count = 1;
    for (hsize_t d : dims) count *= (size_t)d;
    hid_t mt = mem_type(k);
    char *dst = static_cast<char *>(out);
    for (int i = 0; datasets[i]; ++i) {
        std::vector<hsize_t> di = shape_of(f.id, datasets[i]);
        if (di != dims) {
            fail(std::string("shape of ") + datasets[i] + " differs from the first dataset");
            break;
        }
        if (!read_one(f.id, datasets[i], mt, dst)) break;
        dst += count * elem_size(k);
    }
    H5Tclose(mt);
}

// intermediate groups are created as needed (the reference relies on H5LTmake_dataset and
// therefore only writes into existing groups; "/img" at the root is its only use)
void create_dataset(Kind k, const char *name, const char *dataset, int rank, const int *dims, const void *data)
{
    g_err.clear();
    Quiet q;
    File f(name, H5F_ACC_RDWR);
    if (!f.ok() || rank < 0 || (rank && !dims) || !data) {
        if (f.ok()) fail("bad argument");
        return;
    }
    std::vector<hsize_t> hd(rank);
    for (int i = 0; i < rank; ++i) hd[i] = (hsize_t)dims[i];
    hid_t sp = rank ? H5Screate_simple(rank, hd.data(), nullptr) : H5Screate(H5S_SCALAR);
    hid_t lcpl = H5Pcreate(H5P_LINK_CREATE);
    H5Pset_create_intermediate_group(lcpl, 1);
    hid_t mt = mem_type(k);
    hid_t d = H5Dcreate2(f.id, dataset, mt, sp, lcpl, H5P_DEFAULT, H5P_DEFAULT);
    if (d < 0)
        fail(std::string("cannot create ") + dataset);
    else {
        if (H5Dwrite(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0) fail(std::string("cannot write ") + dataset);
        H5Dclose(d);
    }
    H5Tclose(mt);
    H5Pclose(lcpl);
    H5Sclose(sp);
}

herr_t collect(hid_t, const char *name, const H5L_info_t *, void *op)
{
    static_cast<std::vector<std::string> *>(op)->push_back(name);
    return 0;
}

}  // namespace

extern "C" {

const char *h5io_last_error(void) { return g_err.c_str(); }

void createh5File(char *name)
{
    g_err.clear();
    Quiet q;
    hid_t f = H5Fcreate(with_ext(name).c_str(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    if (f < 0)
        fail("cannot create " + with_ext(name));
    else
        H5Fclose(f);
}

int getRankDataset(char *name, char *dataset)
{
    g_err.clear();
    Quiet q;
    File f(name, H5F_ACC_RDONLY);
    if (!f.ok()) return -1;
    hid_t d = H5Dopen2(f.id, dataset, H5P_DEFAULT);
    if (d < 0) {
        fail(std::string("no dataset ") + dataset);
        return -1;
    }
    hid_t sp = H5Dget_space(d);
    int rank = H5Sget_simple_extent_ndims(sp);
    H5Sclose(sp);
    H5Dclose(d);
    return rank;
}

void getDimsDataset(char *name, char *dataset, int rank, int *dims)
{
    g_err.clear();
    Quiet q;
    File f(name, H5F_ACC_RDONLY);
    if (!f.ok() || !dims) return;
    std::vector<hsize_t> s = shape_of(f.id, dataset);
    for (int i = 0; i < rank && i < (int)s.size(); ++i) dims[i] = (int)s[i];
}

void readDatasetInt(char *name, char *dataset, int *data) { read_dataset(K_INT, name, dataset, data); }
void readDatasetLLong(char *name, char *dataset, long long *data) { read_dataset(K_LLONG, name, dataset, data); }
void readDatasetDouble(char *name, char *dataset, double *data) { read_dataset(K_DOUBLE, name, dataset, data); }
void readDatasetComplex(char *name, char *dataset, void *data) { read_dataset(K_COMPLEX, name, dataset, data); }
void readDatasetsDouble(char *name, char **datasets, double *data) { read_datasets(K_DOUBLE, name, datasets, data); }
void readDatasetsComplex(char *name, char **datasets, void *data) { read_datasets(K_COMPLEX, name, datasets, data); }

void createDatasetInt(char *name, char *dataset, int rank, int *dims, int *data)
{
    create_dataset(K_INT, name, dataset, rank, dims, data);
}
void createDatasetLLong(char *name, char *dataset, int rank, int *dims, long long *data)
{
    create_dataset(K_LLONG, name, dataset, rank, dims, data);
}
void createDatasetDouble(char *name, char *dataset, int rank, int *dims, double *data)
{
    create_dataset(K_DOUBLE, name, dataset, rank, dims, data);
}
void createDatasetComplex(char *name, char *dataset, int rank, int *dims, void *data)
{
    create_dataset(K_COMPLEX, name, dataset, rank, dims, data);
}

// malloc'd, NULL-terminated array of malloc'd names (the caller owns it, as with the reference);
// an empty list on failure.
char **listGroupMembers(char *name, char *groupname)
{
    g_err.clear();
    Quiet q;
    std::vector<std::string> names;
    {
        File f(name, H5F_ACC_RDONLY);
        if (f.ok()) {
            hid_t g = H5Gopen2(f.id, groupname, H5P_DEFAULT);
            if (g < 0)
                fail(std::string("no group ") + groupname);
            else {
                H5Literate(g, H5_INDEX_NAME, H5_ITER_NATIVE, nullptr, collect, &names);
                H5Gclose(g);
            }
        }
    }
    char **out = static_cast<char **>(std::malloc((names.size() + 1) * sizeof(char *)));
    if (!out) return nullptr;
    for (size_t i = 0; i < names.size(); ++i) {
        out[i] = static_cast<char *>(std::malloc(names[i].size() + 1));
        std::memcpy(out[i], names[i].c_str(), names[i].size() + 1);
    }
    out[names.size()] = nullptr;
    return out;
}

void h5io_free_list(char **list)
{
    if (!list) return;
    for (char **p = list; *p; ++p) std::free(*p);
    std::free(list);
}

}  // extern "C"
