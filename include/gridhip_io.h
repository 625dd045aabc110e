/*
 * gridhip_io.h — C ABI of libgridhip_io.so: native HDF5 I/O for the gridder's datasets.
 *
 * The entry points are exactly the symbols src/Hdf5.hs:30-67 binds today, i.e. the extern "C"
 * functions of /root/reference/hdf5/hdf5.cc (cited per function), with the same signatures and file
 * conventions: the file is opened and closed inside every call; a name without ".h5" gets it
 * appended (hdf5.cc:343-349, but the caller's buffer is not modified here); complex data is the
 * compound {double r; double i;} (hdf5.cc:14-17,191-211); shapes are `int rank, int *dims` in C
 * order.  Two additions: h5io_last_error() (the reference drops every HDF5 status) and
 * h5io_free_list() for the array listGroupMembers returns.
 */
#ifndef GRIDHIP_IO_H
#define GRIDHIP_IO_H
#ifdef __cplusplus
extern "C" {
#endif

void createh5File(char *name);                                            /* hdf5.cc:59-71  */
int getRankDataset(char *name, char *dataset);                            /* hdf5.cc:124-137 */
void getDimsDataset(char *name, char *dataset, int rank, int *dims);      /* hdf5.cc:139-154 */
void readDatasetInt(char *name, char *dataset, int *data);                /* hdf5.cc:78-81  */
void readDatasetLLong(char *name, char *dataset, long long *data);        /* hdf5.cc:73-76  */
void readDatasetDouble(char *name, char *dataset, double *data);          /* hdf5.cc:83-86  */
void readDatasetComplex(char *name, char *dataset, void *data);           /* hdf5.cc:88-91  */
void readDatasetsDouble(char *name, char **datasets, double *data);       /* hdf5.cc:93-96  (NULL-terminated list) */
void readDatasetsComplex(char *name, char **datasets, void *data);        /* hdf5.cc:98-101 */
void createDatasetInt(char *name, char *dataset, int rank, int *dims, int *data);              /* hdf5.cc:104-107 */
void createDatasetLLong(char *name, char *dataset, int rank, int *dims, long long *data);      /* hdf5.cc:109-112 */
void createDatasetDouble(char *name, char *dataset, int rank, int *dims, double *data);        /* hdf5.cc:114-117 */
void createDatasetComplex(char *name, char *dataset, int rank, int *dims, void *data);         /* hdf5.cc:119-122 */
char **listGroupMembers(char *name, char *groupname);                     /* hdf5.cc:156-186 */

const char *h5io_last_error(void); /* "" when the last call on this thread succeeded */
void h5io_free_list(char **list);

#ifdef __cplusplus
}
#endif
#endif
