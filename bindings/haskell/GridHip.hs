{-# LANGUAGE ForeignFunctionInterface #-}
{-# LANGUAGE ScopedTypeVariables #-}
-- |
-- GridHip: binding of libgridhip.so (include/gridhip.h), the MI355X-native gridders behind the signatures of
-- src/Gridding.hs.  A maintainer of sakehl/SKA-SDP-Accelerate-gridding adds this file as src/GridHip.hs (cabal:
-- other-modules GridHip, extra-libraries gridhip) and replaces the bodies of grid / convgrid / convgrid2 /
-- convgrid3 / convgrid4 as Gridding.patch.md shows; every signature of src/Gridding.hs stays as it is.
--
-- Style and conventions are those of the reference's own binding, src/Hdf5.hs:30-67,113-137:
--   * foreign import ccall, plain pointers and integers, the CALLER allocates outputs (mallocForeignPtrArray here,
--     mallocArray + newForeignPtr finalizerFree there);
--   * Accelerate arrays cross without a copy through accelerate-io (toForeignPtrs / fromForeignPtrs): a
--     Vector (F,F,F) is three buffers, an array of Complex Double ONE interleaved (re,im) buffer
--     (src/Hdf5.hs:113-137 adopts such a buffer as a single ForeignPtr; hdf5/hdf5.cc:14-17 is struct {double r, i;});
--   * shapes travel as plain integers in C order (row-major [y][x] grids, [W][Q][Q][gh][gw] kernel tables).
-- One deliberate departure: every entry point returns a status and `check` turns a failure into `error` with the
-- library's message, where the reference's shim drops HDF5 statuses (hdf5/hdf5.cc:62,70,159).
--
-- STATUS: there is no GHC in the image this library is built and tested in, so this module has never been compiled.
-- The `foreign import` block is generated from include/gridhip.h (bindings/haskell/gen_imports.py) and
-- tests/test_haskell_shim.py checks names, arity and C types of every import against the header; the wrappers below
-- it are written by hand in the idiom of src/Hdf5.hs.
module GridHip
  ( GridHip, withGridHip, openGridHip, closeGridHip, setOption, getOption
  -- * gridders (IO forms of src/Gridding.hs:95-98, 153-157, 199-204, 246-252, 318-324)
  , gridIO, convgridIO, convgrid2IO, degrid2IO, awgridIO
  -- * imaging functions and do_imaging (src/Gridding.hs:76-93, 115-124, 399-449, 452-478, 509-549)
  , simpleImagingIO, convImagingIO, wCacheImagingIO, awImagingIO, doImagingIO, ImagingKind(..)
  -- * a whole node (single process, all devices; RCCL all-reduce of the partial grids)
  , Node, withNode, convgrid2NodeIO
  ) where

import Foreign
import Foreign.C.Types
import Foreign.C.String
import Control.Exception (bracket)
import Control.Monad (when)

import qualified Data.Array.Accelerate                       as A
import qualified Data.Array.Accelerate.Array.Sugar           as A hiding (shape)
import qualified Data.Array.Accelerate.IO.Foreign.ForeignPtr as A
import Data.Array.Accelerate.Data.Complex

import Types   -- F, Visibility, BaseLine, BaseLines, Antenna (src/Types.hs:7-16)

data Ctx
data Plan
data Comm
newtype GridHip = GridHip (Ptr Ctx)
newtype Node    = Node (Ptr Comm)

-- ---------------------------------------------------------------------------------------------------------
-- foreign imports: generated from include/gridhip.h by bindings/haskell/gen_imports.py - do not edit by hand
-- BEGIN GENERATED IMPORTS
-- int gridhip_version()
foreign import ccall unsafe "gridhip_version"
  c_version :: IO CInt
-- const char * gridhip_strerror(code)
foreign import ccall unsafe "gridhip_strerror"
  c_strerror :: CInt -> IO CString
-- int gridhip_device_count(count)
foreign import ccall unsafe "gridhip_device_count"
  c_device_count :: Ptr CInt -> IO CInt
-- int gridhip_create(device, ctx)
foreign import ccall unsafe "gridhip_create"
  c_create :: CInt -> Ptr (Ptr Ctx) -> IO CInt
-- int gridhip_destroy(ctx)
foreign import ccall unsafe "gridhip_destroy"
  c_destroy :: Ptr Ctx -> IO CInt
-- const char * gridhip_last_error(ctx)
foreign import ccall unsafe "gridhip_last_error"
  c_last_error :: Ptr Ctx -> IO CString
-- int gridhip_set_stream(ctx, hip_stream)
foreign import ccall unsafe "gridhip_set_stream"
  c_set_stream :: Ptr Ctx -> Ptr () -> IO CInt
-- int gridhip_reset_stream(ctx)
foreign import ccall unsafe "gridhip_reset_stream"
  c_reset_stream :: Ptr Ctx -> IO CInt
-- void * gridhip_get_stream(ctx)
foreign import ccall unsafe "gridhip_get_stream"
  c_get_stream :: Ptr Ctx -> IO (Ptr ())
-- int gridhip_synchronize(ctx)
foreign import ccall unsafe "gridhip_synchronize"
  c_synchronize :: Ptr Ctx -> IO CInt
-- int gridhip_set_option(ctx, key, value)
foreign import ccall unsafe "gridhip_set_option"
  c_set_option :: Ptr Ctx -> CString -> Int64 -> IO CInt
-- int gridhip_get_option(ctx, key, value)
foreign import ccall unsafe "gridhip_get_option"
  c_get_option :: Ptr Ctx -> CString -> Ptr Int64 -> IO CInt
-- int gridhip_last_dropped(ctx, dropped)
foreign import ccall unsafe "gridhip_last_dropped"
  c_last_dropped :: Ptr Ctx -> Ptr Int64 -> IO CInt
-- int gridhip_grid(ctx, H, Wd, grid, n, u, v, uv_stride, vis)
foreign import ccall unsafe "gridhip_grid"
  c_grid :: Ptr Ctx -> Int64 -> Int64 -> Ptr CDouble -> Int64 -> Ptr CDouble -> Ptr CDouble -> Int64 -> Ptr CDouble -> IO CInt
-- int gridhip_convgrid(ctx, H, Wd, grid, n, Q, gh, gw, gcf, u, v, uv_stride, vis)
foreign import ccall unsafe "gridhip_convgrid"
  c_convgrid :: Ptr Ctx -> Int64 -> Int64 -> Ptr CDouble -> Int64 -> Int64 -> Int64 -> Int64 -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Int64 -> Ptr CDouble -> IO CInt
-- int gridhip_convgrid2(ctx, H, Wd, grid, n, W, Q, gh, gw, gcf, u, v, uv_stride, wbin, vis)
foreign import ccall unsafe "gridhip_convgrid2"
  c_convgrid2 :: Ptr Ctx -> Int64 -> Int64 -> Ptr CDouble -> Int64 -> Int64 -> Int64 -> Int64 -> Int64 -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Int64 -> Ptr Int64 -> Ptr CDouble -> IO CInt
-- int gridhip_degrid2(ctx, H, Wd, grid, n, W, Q, gh, gw, gcf, u, v, uv_stride, wbin, vis_out)
foreign import ccall unsafe "gridhip_degrid2"
  c_degrid2 :: Ptr Ctx -> Int64 -> Int64 -> Ptr CDouble -> Int64 -> Int64 -> Int64 -> Int64 -> Int64 -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Int64 -> Ptr Int64 -> Ptr CDouble -> IO CInt
-- int gridhip_grid_dev(ctx, H, Wd, grid, n, u, v, uv_stride, vis)
foreign import ccall unsafe "gridhip_grid_dev"
  c_grid_dev :: Ptr Ctx -> Int64 -> Int64 -> Ptr CDouble -> Int64 -> Ptr CDouble -> Ptr CDouble -> Int64 -> Ptr CDouble -> IO CInt
-- int gridhip_convgrid_dev(ctx, H, Wd, grid, n, Q, gh, gw, gcf, u, v, uv_stride, vis)
foreign import ccall unsafe "gridhip_convgrid_dev"
  c_convgrid_dev :: Ptr Ctx -> Int64 -> Int64 -> Ptr CDouble -> Int64 -> Int64 -> Int64 -> Int64 -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Int64 -> Ptr CDouble -> IO CInt
-- int gridhip_convgrid2_dev(ctx, H, Wd, grid, n, W, Q, gh, gw, gcf, u, v, uv_stride, wbin, vis)
foreign import ccall unsafe "gridhip_convgrid2_dev"
  c_convgrid2_dev :: Ptr Ctx -> Int64 -> Int64 -> Ptr CDouble -> Int64 -> Int64 -> Int64 -> Int64 -> Int64 -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Int64 -> Ptr Int64 -> Ptr CDouble -> IO CInt
-- int gridhip_degrid2_dev(ctx, H, Wd, grid, n, W, Q, gh, gw, gcf, u, v, uv_stride, wbin, vis_out)
foreign import ccall unsafe "gridhip_degrid2_dev"
  c_degrid2_dev :: Ptr Ctx -> Int64 -> Int64 -> Ptr CDouble -> Int64 -> Int64 -> Int64 -> Int64 -> Int64 -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Int64 -> Ptr Int64 -> Ptr CDouble -> IO CInt
-- int gridhip_plan_create_dev(ctx, H, Wd, n, W, Q, gh, gw, u, v, uv_stride, wbin, plan)
foreign import ccall unsafe "gridhip_plan_create_dev"
  c_plan_create_dev :: Ptr Ctx -> Int64 -> Int64 -> Int64 -> Int64 -> Int64 -> Int64 -> Int64 -> Ptr CDouble -> Ptr CDouble -> Int64 -> Ptr Int64 -> Ptr (Ptr Plan) -> IO CInt
-- int gridhip_plan_grid_dev(plan, gcf, vis, grid)
foreign import ccall unsafe "gridhip_plan_grid_dev"
  c_plan_grid_dev :: Ptr Plan -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> IO CInt
-- int gridhip_plan_degrid_dev(plan, gcf, grid, vis_out)
foreign import ccall unsafe "gridhip_plan_degrid_dev"
  c_plan_degrid_dev :: Ptr Plan -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> IO CInt
-- int gridhip_plan_destroy(plan)
foreign import ccall unsafe "gridhip_plan_destroy"
  c_plan_destroy :: Ptr Plan -> IO CInt
-- int64_t gridhip_image_size(theta, lam)
foreign import ccall unsafe "gridhip_image_size"
  c_image_size :: CDouble -> Int64 -> IO Int64
-- int gridhip_wbins(ctx, n, w, wstep, wbin, wmin, nplanes)
foreign import ccall unsafe "gridhip_wbins"
  c_wbins :: Ptr Ctx -> Int64 -> Ptr CDouble -> Int64 -> Ptr Int64 -> Ptr Int64 -> Ptr Int64 -> IO CInt
-- int gridhip_find_closest(ctx, nws, ws, n, w, out)
foreign import ccall unsafe "gridhip_find_closest"
  c_find_closest :: Ptr Ctx -> Int64 -> Ptr CDouble -> Int64 -> Ptr CDouble -> Ptr Int64 -> IO CInt
-- int gridhip_mirror_uvw(ctx, n, u, v, w, vis)
foreign import ccall unsafe "gridhip_mirror_uvw"
  c_mirror_uvw :: Ptr Ctx -> Int64 -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> IO CInt
-- int gridhip_doweight(ctx, theta, lam, n, u, v, vis)
foreign import ccall unsafe "gridhip_doweight"
  c_doweight :: Ptr Ctx -> CDouble -> Int64 -> Int64 -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> IO CInt
-- int gridhip_make_grid_hermitian(ctx, N, grid)
foreign import ccall unsafe "gridhip_make_grid_hermitian"
  c_make_grid_hermitian :: Ptr Ctx -> Int64 -> Ptr CDouble -> IO CInt
-- int gridhip_fft2_centered(ctx, N, in, out, inverse)
foreign import ccall unsafe "gridhip_fft2_centered"
  c_fft2_centered :: Ptr Ctx -> Int64 -> Ptr CDouble -> Ptr CDouble -> CInt -> IO CInt
-- int gridhip_w_kernel(ctx, theta, w, npixFF, npixKern, qpx, out)
foreign import ccall unsafe "gridhip_w_kernel"
  c_w_kernel :: Ptr Ctx -> CDouble -> CDouble -> Int64 -> Int64 -> Int64 -> Ptr CDouble -> IO CInt
-- int gridhip_simple_imaging(ctx, theta, lam, n, u, v, uv_stride, vis, grid)
foreign import ccall unsafe "gridhip_simple_imaging"
  c_simple_imaging :: Ptr Ctx -> CDouble -> Int64 -> Int64 -> Ptr CDouble -> Ptr CDouble -> Int64 -> Ptr CDouble -> Ptr CDouble -> IO CInt
-- int gridhip_conv_imaging(ctx, Q, gh, gw, kv, theta, lam, n, u, v, uv_stride, vis, grid)
foreign import ccall unsafe "gridhip_conv_imaging"
  c_conv_imaging :: Ptr Ctx -> Int64 -> Int64 -> Int64 -> Ptr CDouble -> CDouble -> Int64 -> Int64 -> Ptr CDouble -> Ptr CDouble -> Int64 -> Ptr CDouble -> Ptr CDouble -> IO CInt
-- int gridhip_w_cache_imaging(ctx, wstep, qpx, npixFF, npixKern, theta, lam, n, u, v, w, uv_stride, vis, grid)
foreign import ccall unsafe "gridhip_w_cache_imaging"
  c_w_cache_imaging :: Ptr Ctx -> Int64 -> Int64 -> Int64 -> Int64 -> CDouble -> Int64 -> Int64 -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Int64 -> Ptr CDouble -> Ptr CDouble -> IO CInt
-- int gridhip_awgrid(ctx, H, Wd, grid, n, W, Q, S, A, wkerns, akerns, u, v, uv_stride, wbin, a1, a2, vis)
foreign import ccall unsafe "gridhip_awgrid"
  c_awgrid :: Ptr Ctx -> Int64 -> Int64 -> Ptr CDouble -> Int64 -> Int64 -> Int64 -> Int64 -> Int64 -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Int64 -> Ptr Int64 -> Ptr Int64 -> Ptr Int64 -> Ptr CDouble -> IO CInt
-- int gridhip_awgrid_dev(ctx, H, Wd, grid, n, W, Q, S, A, wkerns, akerns, u, v, uv_stride, wbin, a1, a2, vis)
foreign import ccall unsafe "gridhip_awgrid_dev"
  c_awgrid_dev :: Ptr Ctx -> Int64 -> Int64 -> Ptr CDouble -> Int64 -> Int64 -> Int64 -> Int64 -> Int64 -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Int64 -> Ptr Int64 -> Ptr Int64 -> Ptr Int64 -> Ptr CDouble -> IO CInt
-- int gridhip_aw_last_stats(ctx, vis_keyed, kernels_built)
foreign import ccall unsafe "gridhip_aw_last_stats"
  c_aw_last_stats :: Ptr Ctx -> Ptr Int64 -> Ptr Int64 -> IO CInt
-- int gridhip_aw_imaging(ctx, theta, lam, W, Q, S, A, wkerns, wvals, akerns, n, u, v, w, uv_stride, a1, a2, vis, grid)
foreign import ccall unsafe "gridhip_aw_imaging"
  c_aw_imaging :: Ptr Ctx -> CDouble -> Int64 -> Int64 -> Int64 -> Int64 -> Int64 -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Int64 -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Int64 -> Ptr Int64 -> Ptr Int64 -> Ptr CDouble -> Ptr CDouble -> IO CInt
-- int gridhip_do_imaging(ctx, kind, wstep, Q, npixFF, gh, gw, kv, theta, lam, n, u, v, w, uv_stride, vis, image, psf, pmax)
foreign import ccall unsafe "gridhip_do_imaging"
  c_do_imaging :: Ptr Ctx -> CInt -> Int64 -> Int64 -> Int64 -> Int64 -> Int64 -> Ptr CDouble -> CDouble -> Int64 -> Int64 -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Int64 -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> IO CInt
-- int gridhip_do_imaging_dev(ctx, kind, wstep, Q, npixFF, gh, gw, kv, theta, lam, n, u, v, w, uv_stride, vis, image, psf, pmax)
foreign import ccall unsafe "gridhip_do_imaging_dev"
  c_do_imaging_dev :: Ptr Ctx -> CInt -> Int64 -> Int64 -> Int64 -> Int64 -> Int64 -> Ptr CDouble -> CDouble -> Int64 -> Int64 -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Int64 -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> IO CInt
-- int gridhip_w_cache_imaging_dev(ctx, wstep, qpx, npixFF, npixKern, theta, lam, n, u, v, w, uv_stride, vis, grid)
foreign import ccall unsafe "gridhip_w_cache_imaging_dev"
  c_w_cache_imaging_dev :: Ptr Ctx -> Int64 -> Int64 -> Int64 -> Int64 -> CDouble -> Int64 -> Int64 -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Int64 -> Ptr CDouble -> Ptr CDouble -> IO CInt
-- int gridhip_comm_create(ndev, dev_ids, comm)
foreign import ccall safe "gridhip_comm_create"
  c_comm_create :: CInt -> Ptr CInt -> Ptr (Ptr Comm) -> IO CInt
-- int gridhip_comm_unique_id(id128)
foreign import ccall unsafe "gridhip_comm_unique_id"
  c_comm_unique_id :: Ptr () -> IO CInt
-- int gridhip_comm_create_rank(ctx, nranks, rank, id128, comm)
foreign import ccall safe "gridhip_comm_create_rank"
  c_comm_create_rank :: Ptr Ctx -> CInt -> CInt -> Ptr () -> Ptr (Ptr Comm) -> IO CInt
-- int gridhip_comm_destroy(comm)
foreign import ccall safe "gridhip_comm_destroy"
  c_comm_destroy :: Ptr Comm -> IO CInt
-- const char * gridhip_comm_last_error(comm)
foreign import ccall unsafe "gridhip_comm_last_error"
  c_comm_last_error :: Ptr Comm -> IO CString
-- int gridhip_comm_ndev(comm)
foreign import ccall unsafe "gridhip_comm_ndev"
  c_comm_ndev :: Ptr Comm -> IO CInt
-- int gridhip_comm_nranks(comm)
foreign import ccall unsafe "gridhip_comm_nranks"
  c_comm_nranks :: Ptr Comm -> IO CInt
-- gridhip_ctx * gridhip_comm_ctx(comm, i)
foreign import ccall unsafe "gridhip_comm_ctx"
  c_comm_ctx :: Ptr Comm -> CInt -> IO (Ptr Ctx)
-- int gridhip_comm_allreduce_grids(comm, cells, grids)
foreign import ccall unsafe "gridhip_comm_allreduce_grids"
  c_comm_allreduce_grids :: Ptr Comm -> Int64 -> Ptr (Ptr CDouble) -> IO CInt
-- int gridhip_comm_allreduce_grid(comm, cells, grid)
foreign import ccall unsafe "gridhip_comm_allreduce_grid"
  c_comm_allreduce_grid :: Ptr Comm -> Int64 -> Ptr CDouble -> IO CInt
-- int gridhip_comm_allreduce_rows(comm, Wd, y0, y1, grids)
foreign import ccall unsafe "gridhip_comm_allreduce_rows"
  c_comm_allreduce_rows :: Ptr Comm -> Int64 -> Int64 -> Int64 -> Ptr (Ptr CDouble) -> IO CInt
-- int gridhip_comm_allreduce_grid_rows(comm, Wd, y0, y1, grid)
foreign import ccall unsafe "gridhip_comm_allreduce_grid_rows"
  c_comm_allreduce_grid_rows :: Ptr Comm -> Int64 -> Int64 -> Int64 -> Ptr CDouble -> IO CInt
-- int gridhip_comm_set_option(comm, key, value)
foreign import ccall unsafe "gridhip_comm_set_option"
  c_comm_set_option :: Ptr Comm -> CString -> Int64 -> IO CInt
-- int gridhip_comm_get_option(comm, key, value)
foreign import ccall unsafe "gridhip_comm_get_option"
  c_comm_get_option :: Ptr Comm -> CString -> Ptr Int64 -> IO CInt
-- int gridhip_comm_set_stream(comm, i, hip_stream)
foreign import ccall unsafe "gridhip_comm_set_stream"
  c_comm_set_stream :: Ptr Comm -> CInt -> Ptr () -> IO CInt
-- int gridhip_comm_reset_stream(comm, i)
foreign import ccall unsafe "gridhip_comm_reset_stream"
  c_comm_reset_stream :: Ptr Comm -> CInt -> IO CInt
-- int gridhip_comm_convgrid2(comm, H, Wd, grid, n, W, Q, gh, gw, gcf, u, v, uv_stride, wbin, vis)
foreign import ccall safe "gridhip_comm_convgrid2"
  c_comm_convgrid2 :: Ptr Comm -> Int64 -> Int64 -> Ptr CDouble -> Int64 -> Int64 -> Int64 -> Int64 -> Int64 -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> Int64 -> Ptr Int64 -> Ptr CDouble -> IO CInt
-- int gridhip_malloc(ctx, dptr, bytes)
foreign import ccall unsafe "gridhip_malloc"
  c_malloc :: Ptr Ctx -> Ptr (Ptr ()) -> Int64 -> IO CInt
-- int gridhip_free(ctx, dptr)
foreign import ccall unsafe "gridhip_free"
  c_free :: Ptr Ctx -> Ptr () -> IO CInt
-- int gridhip_memcpy_h2d(ctx, dst, src, bytes)
foreign import ccall unsafe "gridhip_memcpy_h2d"
  c_memcpy_h2d :: Ptr Ctx -> Ptr () -> Ptr () -> Int64 -> IO CInt
-- int gridhip_memcpy_d2h(ctx, dst, src, bytes)
foreign import ccall unsafe "gridhip_memcpy_d2h"
  c_memcpy_d2h :: Ptr Ctx -> Ptr () -> Ptr () -> Int64 -> IO CInt
-- int gridhip_memset(ctx, dptr, value, bytes)
foreign import ccall unsafe "gridhip_memset"
  c_memset :: Ptr Ctx -> Ptr () -> CInt -> Int64 -> IO CInt
-- int gridhip_last_timing(ctx, ms_total, ms_prepass, ms_kernel)
foreign import ccall unsafe "gridhip_last_timing"
  c_last_timing :: Ptr Ctx -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> IO CInt
-- int gridhip_timing(ctx, back, ms_total, ms_prepass, ms_kernel)
foreign import ccall unsafe "gridhip_timing"
  c_timing :: Ptr Ctx -> CInt -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> IO CInt
-- int gridhip_enable_timing(ctx, enable)
foreign import ccall unsafe "gridhip_enable_timing"
  c_enable_timing :: Ptr Ctx -> CInt -> IO CInt
-- END GENERATED IMPORTS
-- ---------------------------------------------------------------------------------------------------------

fi :: (Integral a, Num b) => a -> b
fi = fromIntegral

-- | a context on HIP device `dev` (one device, one stream; not thread-safe: one per calling thread)
openGridHip :: Int -> IO GridHip
openGridHip dev = alloca $ \pp -> do
  rc <- c_create (fi dev) pp
  when (rc /= 0) $ c_strerror rc >>= peekCString >>= \m -> error ("gridhip_create: " ++ m)
  GridHip <$> peek pp

closeGridHip :: GridHip -> IO ()
closeGridHip (GridHip p) = () <$ c_destroy p

withGridHip :: Int -> (GridHip -> IO a) -> IO a
withGridHip dev = bracket (openGridHip dev) closeGridHip

check :: GridHip -> CInt -> IO ()
check (GridHip p) rc = when (rc /= 0) $ do
  msg <- c_last_error p >>= peekCString
  error ("gridhip (" ++ show rc ++ "): " ++ msg)

setOption :: GridHip -> String -> Int -> IO ()
setOption h@(GridHip p) key val = withCString key $ \k -> c_set_option p k (fi val) >>= check h

getOption :: GridHip -> String -> IO Int
getOption h@(GridHip p) key = withCString key $ \k -> alloca $ \o -> do
  c_get_option p k o >>= check h
  fi <$> peek o

-- Pointer views of Accelerate arrays (no copy).  Complex Double arrays are one interleaved buffer of doubles; a
-- Vector (F,F,F) is three buffers, a Vector (Int,Int,Int) likewise.
withCplx :: A.Shape sh => A.Array sh Visibility -> (Ptr CDouble -> IO b) -> IO b
withCplx arr k = withForeignPtr (castForeignPtr (A.toForeignPtrs arr)) k

withF :: A.Shape sh => A.Array sh F -> (Ptr CDouble -> IO b) -> IO b
withF arr k = withForeignPtr (castForeignPtr (A.toForeignPtrs arr)) k

withI :: A.Shape sh => A.Array sh Int -> (Ptr Int64 -> IO b) -> IO b
withI arr k = withForeignPtr (castForeignPtr (A.toForeignPtrs arr)) k

withI64 :: A.Shape sh => A.Array sh Antenna -> (Ptr Int64 -> IO b) -> IO b
withI64 arr k = withForeignPtr (castForeignPtr (A.toForeignPtrs arr)) k

withUVW :: A.Vector BaseLines -> (Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> IO b) -> IO b
withUVW p k =
  let ((((), pu), pv), pw) = A.toForeignPtrs p
  in withForeignPtr (castForeignPtr pu) $ \u -> withForeignPtr (castForeignPtr pv) $ \v ->
     withForeignPtr (castForeignPtr pw) $ \w -> k u v w

withIdx3 :: A.Vector (Int, Int, Int) -> (Ptr Int64 -> Ptr Int64 -> Ptr Int64 -> IO b) -> IO b
withIdx3 ix k =
  let ((((), p0), p1), p2) = A.toForeignPtrs ix
  in withForeignPtr (castForeignPtr p0) $ \a -> withForeignPtr (castForeignPtr p1) $ \b ->
     withForeignPtr (castForeignPtr p2) $ \c -> k a b c

-- a fresh copy of the destination grid: every gridder ACCUMULATES INTO the grid it is given, as
-- `permute (+) a ...` does (src/Gridding.hs:99,197,244), and Accelerate arrays are immutable
copyGrid :: A.Matrix Visibility -> IO (ForeignPtr CDouble, Int, Int)
copyGrid a = do
  let A.Z A.:. hgt A.:. wid = A.arrayShape a
  out <- mallocForeignPtrArray (2 * hgt * wid)
  withForeignPtr out $ \o -> withCplx a $ \src -> copyArray o src (2 * hgt * wid)
  return (out, hgt, wid)

adoptGrid :: ForeignPtr CDouble -> Int -> Int -> A.Matrix Visibility
adoptGrid out hgt wid = A.fromForeignPtrs (A.Z A.:. hgt A.:. wid) (castForeignPtr out)

newGrid :: Int -> IO (ForeignPtr CDouble)
newGrid n = mallocForeignPtrArray (2 * n * n)

-- | grid a p v  (src/Gridding.hs:95-98)
gridIO :: GridHip -> A.Matrix Visibility -> A.Vector BaseLines -> A.Vector Visibility -> IO (A.Matrix Visibility)
gridIO h@(GridHip c) a p v = do
  (out, hgt, wid) <- copyGrid a
  let A.Z A.:. n = A.arrayShape v
  withForeignPtr out $ \o -> withUVW p $ \pu pv _ -> withCplx v $ \vs ->
    c_grid c (fi hgt) (fi wid) o (fi n) pu pv 1 vs >>= check h
  return (adoptGrid out hgt wid)

-- | convgrid gcf a p v  (src/Gridding.hs:153-157); gcf is [Q,Q,gh,gw]
convgridIO :: GridHip -> A.Array A.DIM4 Visibility -> A.Matrix Visibility -> A.Vector BaseLines
           -> A.Vector Visibility -> IO (A.Matrix Visibility)
convgridIO h@(GridHip c) gcf a p v = do
  (out, hgt, wid) <- copyGrid a
  let A.Z A.:. q A.:. _ A.:. gh A.:. gw = A.arrayShape gcf
      A.Z A.:. n = A.arrayShape v
  withForeignPtr out $ \o -> withCplx gcf $ \k -> withUVW p $ \pu pv _ -> withCplx v $ \vs ->
    c_convgrid c (fi hgt) (fi wid) o (fi n) (fi q) (fi gh) (fi gw) k pu pv 1 vs >>= check h
  return (adoptGrid out hgt wid)

-- | convgrid2 gcf a p wbin v  (src/Gridding.hs:199-204); gcf is [W,Q,Q,gh,gw]
convgrid2IO :: GridHip -> A.Array A.DIM5 Visibility -> A.Matrix Visibility -> A.Vector BaseLines
            -> A.Vector Int -> A.Vector Visibility -> IO (A.Matrix Visibility)
convgrid2IO h@(GridHip c) gcf a p wbin v = do
  (out, hgt, wid) <- copyGrid a
  let A.Z A.:. w A.:. q A.:. _ A.:. gh A.:. gw = A.arrayShape gcf
      A.Z A.:. n = A.arrayShape v
  withForeignPtr out $ \o -> withCplx gcf $ \k -> withUVW p $ \pu pv _ -> withI wbin $ \wb -> withCplx v $ \vs ->
    c_convgrid2 c (fi hgt) (fi wid) o (fi n) (fi w) (fi q) (fi gh) (fi gw) k pu pv 1 wb vs >>= check h
  return (adoptGrid out hgt wid)

-- | degrid2 gcf a p wbin: the gather with convgrid2's coordinates (north star "degrid"; the reference has none)
degrid2IO :: GridHip -> A.Array A.DIM5 Visibility -> A.Matrix Visibility -> A.Vector BaseLines
          -> A.Vector Int -> IO (A.Vector Visibility)
degrid2IO h@(GridHip c) gcf a p wbin = do
  let A.Z A.:. w A.:. q A.:. _ A.:. gh A.:. gw = A.arrayShape gcf
      A.Z A.:. hgt A.:. wid = A.arrayShape a
      A.Z A.:. n = A.arrayShape wbin
  out <- mallocForeignPtrArray (2 * n) :: IO (ForeignPtr CDouble)
  withForeignPtr out $ \o -> withCplx gcf $ \k -> withCplx a $ \g -> withUVW p $ \pu pv _ -> withI wbin $ \wb ->
    c_degrid2 c (fi hgt) (fi wid) g (fi n) (fi w) (fi q) (fi gh) (fi gw) k pu pv 1 wb o >>= check h
  return (A.fromForeignPtrs (A.Z A.:. n) (castForeignPtr out))

-- | convgrid3 / convgrid4 wkerns akerns a p index v  (src/Gridding.hs:246-252, 318-324: both produce this grid);
-- wkerns [W,Q,Q,S,S], akerns [A,S,S], index = (wbin, a1, a2)
awgridIO :: GridHip -> A.Array A.DIM5 Visibility -> A.Array A.DIM3 Visibility -> A.Matrix Visibility
         -> A.Vector BaseLines -> A.Vector (Int, Int, Int) -> A.Vector Visibility -> IO (A.Matrix Visibility)
awgridIO h@(GridHip c) wkerns akerns a p index v = do
  (out, hgt, wid) <- copyGrid a
  let A.Z A.:. w A.:. q A.:. _ A.:. s A.:. _ = A.arrayShape wkerns
      A.Z A.:. na A.:. _ A.:. _ = A.arrayShape akerns
      A.Z A.:. n = A.arrayShape v
  withForeignPtr out $ \o -> withCplx wkerns $ \wk -> withCplx akerns $ \ak -> withUVW p $ \pu pv _ ->
    withIdx3 index $ \wb a1 a2 -> withCplx v $ \vs ->
      c_awgrid c (fi hgt) (fi wid) o (fi n) (fi w) (fi q) (fi s) (fi na) wk ak pu pv 1 wb a1 a2 vs >>= check h
  return (adoptGrid out hgt wid)

-- ---------------------------------------------------------------------------------------------------------
-- ImagingFunctions (src/Gridding.hs:76-81): theta lam uvw src vis -> grid.  uvw in wavelengths; `src` is only used
-- by the aw variants (:475), which take the antenna vectors directly here.

imageSize :: F -> Int -> IO Int
imageSize theta lam = fi <$> c_image_size (realToFrac theta) (fi lam)

-- | simple_imaging  (src/Gridding.hs:84-93)
simpleImagingIO :: GridHip -> F -> Int -> A.Vector BaseLines -> A.Vector Visibility -> IO (A.Matrix Visibility)
simpleImagingIO h@(GridHip c) theta lam uvw vis = do
  n' <- imageSize theta lam
  out <- newGrid n'
  let A.Z A.:. n = A.arrayShape vis
  withForeignPtr out $ \o -> withUVW uvw $ \u v _ -> withCplx vis $ \vs ->
    c_simple_imaging c (realToFrac theta) (fi lam) (fi n) u v 1 vs o >>= check h
  return (adoptGrid out n' n')

-- | conv_imaging kv  (src/Gridding.hs:115-124)
convImagingIO :: GridHip -> A.Array A.DIM4 Visibility -> F -> Int -> A.Vector BaseLines -> A.Vector Visibility
              -> IO (A.Matrix Visibility)
convImagingIO h@(GridHip c) kv theta lam uvw vis = do
  n' <- imageSize theta lam
  out <- newGrid n'
  let A.Z A.:. q A.:. _ A.:. gh A.:. gw = A.arrayShape kv
      A.Z A.:. n = A.arrayShape vis
  withForeignPtr out $ \o -> withCplx kv $ \k -> withUVW uvw $ \u v _ -> withCplx vis $ \vs ->
    c_conv_imaging c (fi q) (fi gh) (fi gw) k (realToFrac theta) (fi lam) (fi n) u v 1 vs o >>= check h
  return (adoptGrid out n' n')

-- | w_cache_imaging kernops otargs  (src/Gridding.hs:399-449): wstep, qpx, npixFF, npixKern of KernelOptions (:30-38)
wCacheImagingIO :: GridHip -> Int -> Int -> Int -> Int -> F -> Int -> A.Vector BaseLines -> A.Vector Visibility
                -> IO (A.Matrix Visibility)
wCacheImagingIO h@(GridHip c) wstep qpx npixFF npixKern theta lam uvw vis = do
  n' <- imageSize theta lam
  out <- newGrid n'
  let A.Z A.:. n = A.arrayShape vis
  withForeignPtr out $ \o -> withUVW uvw $ \u v w -> withCplx vis $ \vs ->
    c_w_cache_imaging c (fi wstep) (fi qpx) (fi npixFF) (fi npixKern) (realToFrac theta) (fi lam) (fi n) u v w 1 vs o
      >>= check h
  return (adoptGrid out n' n')

-- | aw_imaging / aw_imagingOld  (src/Gridding.hs:452-506); wvals = the W plane w-values findClosest searches
awImagingIO :: GridHip -> F -> Int -> A.Array A.DIM5 Visibility -> A.Vector BaseLine -> A.Array A.DIM3 Visibility
            -> A.Vector BaseLines -> A.Vector Antenna -> A.Vector Antenna -> A.Vector Visibility
            -> IO (A.Matrix Visibility)
awImagingIO h@(GridHip c) theta lam wkerns wvals akerns uvw ant1 ant2 vis = do
  n' <- imageSize theta lam
  out <- newGrid n'
  let A.Z A.:. w A.:. q A.:. _ A.:. s A.:. _ = A.arrayShape wkerns
      A.Z A.:. na A.:. _ A.:. _ = A.arrayShape akerns
      A.Z A.:. n = A.arrayShape vis
  withForeignPtr out $ \o -> withCplx wkerns $ \wk -> withF wvals $ \wv -> withCplx akerns $ \ak ->
    withUVW uvw $ \u v ww -> withI64 ant1 $ \a1 -> withI64 ant2 $ \a2 -> withCplx vis $ \vs ->
      c_aw_imaging c (realToFrac theta) (fi lam) (fi w) (fi q) (fi s) (fi na) wk wv ak (fi n) u v ww 1 a1 a2 vs o
        >>= check h
  return (adoptGrid out n' n')

-- | which ImagingFunction do_imaging runs (the `imgfn` argument of src/Gridding.hs:509-519)
data ImagingKind
  = SimpleImaging                              -- ^ simple_imaging
  | ConvImaging (A.Array A.DIM4 Visibility)    -- ^ conv_imaging kv
  | WCacheImaging Int Int Int Int              -- ^ w_cache_imaging: wstep qpx npixFF npixKern

-- | do_imaging theta lam uvw a1 a2 t f vis imgfn  (src/Gridding.hs:509-549): (image, psf, pmax).
-- uvw is the (n,3) row-major Matrix BaseLine as it comes from HDF5 (src/ImageDataset.hs:94-97): passed with
-- uv_stride = 3, the columns are sliced on the device (:524-526).  a1, a2, t, f are unused by these imaging functions.
doImagingIO :: GridHip -> F -> Int -> A.Matrix BaseLine -> A.Vector Visibility -> ImagingKind
            -> IO (A.Matrix F, A.Matrix F, F)
doImagingIO h@(GridHip c) theta lam uvw vis kind = do
  n' <- imageSize theta lam
  img <- mallocForeignPtrArray (n' * n') :: IO (ForeignPtr CDouble)
  psf <- mallocForeignPtrArray (n' * n') :: IO (ForeignPtr CDouble)
  let A.Z A.:. n = A.arrayShape vis
      run k wstep q npixFF gh gw kv =
        withForeignPtr img $ \pi' -> withForeignPtr psf $ \pp -> withF uvw $ \m -> withCplx vis $ \vs ->
          alloca $ \pm -> do
            c_do_imaging c k (fi wstep) (fi q) (fi npixFF) (fi gh) (fi gw) kv (realToFrac theta) (fi lam) (fi n)
                         m (m `advancePtr` 1) (m `advancePtr` 2) 3 vs pi' pp pm >>= check h
            realToFrac <$> peek pm
  pmax <- case kind of
    SimpleImaging -> run 0 (0 :: Int) (0 :: Int) (0 :: Int) (0 :: Int) (0 :: Int) nullPtr
    ConvImaging kv ->
      let A.Z A.:. q A.:. _ A.:. gh A.:. gw = A.arrayShape kv
      in withCplx kv $ \k -> run 1 (0 :: Int) q (0 :: Int) gh gw k
    WCacheImaging wstep q npixFF s -> run 2 wstep q npixFF s s nullPtr
  let sh = A.Z A.:. n' A.:. n'
  return (A.fromForeignPtrs sh (castForeignPtr img), A.fromForeignPtrs sh (castForeignPtr psf), pmax)

-- ---------------------------------------------------------------------------------------------------------
-- A whole node from one Haskell process: ndev devices, visibilities cut into contiguous shards, partial grids
-- summed with one RCCL all-reduce over xGMI (include/gridhip.h, gridhip_comm_*).

withNode :: Int -> (Node -> IO a) -> IO a
withNode ndev = bracket open (\(Node c) -> () <$ c_comm_destroy c)
  where open = alloca $ \pp -> do
                 rc <- c_comm_create (fi ndev) nullPtr pp
                 when (rc /= 0) $ c_comm_last_error nullPtr >>= peekCString >>= error
                 Node <$> peek pp

-- | convgrid2 over all devices of the node (same arguments and result as convgrid2IO)
convgrid2NodeIO :: Node -> A.Array A.DIM5 Visibility -> A.Matrix Visibility -> A.Vector BaseLines
                -> A.Vector Int -> A.Vector Visibility -> IO (A.Matrix Visibility)
convgrid2NodeIO (Node c) gcf a p wbin v = do
  (out, hgt, wid) <- copyGrid a
  let A.Z A.:. w A.:. q A.:. _ A.:. gh A.:. gw = A.arrayShape gcf
      A.Z A.:. n = A.arrayShape v
  rc <- withForeignPtr out $ \o -> withCplx gcf $ \k -> withUVW p $ \pu pv _ -> withI wbin $ \wb -> withCplx v $ \vs ->
          c_comm_convgrid2 c (fi hgt) (fi wid) o (fi n) (fi w) (fi q) (fi gh) (fi gw) k pu pv 1 wb vs
  when (rc /= 0) $ c_comm_last_error c >>= peekCString >>= \m -> error ("gridhip node: " ++ m)
  return (adoptGrid out hgt wid)
