-- integration/LibHip.hs -- the Haskell side of the drop-in: what a maintainer of rrruko/squigly-trace adds as
-- src/LibHip.hs to run Lib.render's parallel fan-out (src/Lib.hs:73-74) on an MI355X through libsquigly_hip.so.
-- Shipped as source only: neither the build container nor the GPU box of this repository has GHC / stack / cabal
-- (DESIGN.md section 3), so it has never been compiled here.  The same C entry points are exercised through ctypes by
-- every GPU test (squigly-trace_amd/_native.py).  How to wire it in: INTEGRATION.md section 2.
--
--   squigly-trace.cabal:  exposed-modules: ... LibHip ; extra-libraries: squigly_hip ; include-dirs / extra-lib-dirs
--   app/Main.hs:43     :  renderHip scene cam settings     (instead of render scene cam settings)
{-# LANGUAGE ForeignFunctionInterface #-}
-- src/LibHip.hs : the MI355X drop-in for Lib.render's parallel fan-out (src/Lib.hs:73-74)
module LibHip (renderHip) where

import           BIH
import           Color
import           Geometry
import           Lib                          (Settings (..), SavePath (..))
import           V3
import           Data.Massiv.Array            (Array, S (..), Ix2 (..), Comp (..))
import qualified Data.Massiv.Array.Unsafe     as A
import           Data.Massiv.Array.IO         (writeImage)
import qualified Data.Matrix                  as M
import qualified Data.Vector                  as V
import qualified Data.Vector.Storable         as VS
import qualified Data.Vector.Storable.Mutable as VSM
import           Data.Int                     (Int32)
import           Data.Word                    (Word8)
import           Foreign
import           Foreign.C.String             (peekCString, CString)
import           Foreign.C.Types

-- typedef struct { int32_t kind; float lmax, rmin; int32_t link; } sq_node;      (pre-order)
-- typedef struct { float v0[3], v1[3], v2[3]; int32_t mat; } sq_tri;             (BIH.flatten order)
-- typedef struct { float reflective, surf[3], emissive, emit[3]; } sq_material;
-- typedef struct { float pos[3]; float rot[9]; } sq_camera;
-- typedef struct { sq_bounds root; const sq_node* nodes; int32_t n_nodes; const sq_tri* tris; int32_t n_tris;
--                  const sq_material* mats; int32_t n_mats; int32_t height; } sq_scene;
data SqScene   -- opaque to Haskell: filled with pokeByteOff below
data SqCamera

-- `safe`: the call runs for seconds and must not block the RTS capability.
foreign import ccall safe "squigly_hip.h sq_render_rgb8"
  c_sq_render_rgb8 :: Ptr SqScene -> Ptr SqCamera -> Int32 -> Int32 -> Int32 -> Int32 -> Ptr Word8 -> IO CInt
foreign import ccall unsafe "squigly_hip.h sq_last_error"
  c_sq_last_error :: IO CString

-- Flatten `Tree BIHNode (Vector Triangle)` (src/BIH.hs:26,37-40) to pre-order sq_node records,
-- triangles in BIH.flatten order (src/BIH.hs:50-52), materials de-duplicated by index.
flattenBIH :: BIHTree -> ([(Int32, Float, Float, Int32)], [Triangle])
flattenBIH t = go t 0 0
  where
    go (Leaf ts) _ firstTri =
      ([(3 + 4 * fromIntegral (V.length ts), 0, 0, fromIntegral firstTri)], V.toList ts)
    go (Branch (BIHN ax lmax rmin) l r) me firstTri =
      let (ln, lt) = go l (me + 1) firstTri
          rightIx  = me + 1 + length ln
          (rn, rt) = go r rightIx (firstTri + length lt)
          axis X = 0; axis Y = 1; axis Z = 2
      in  ((axis ax, lmax, rmin, fromIntegral rightIx) : ln ++ rn, lt ++ rt)

-- | Drop-in for the body of Lib.render (src/Lib.hs:68-75).
renderHip :: Scene BIH -> Camera -> Settings -> IO ()
renderHip (Scene bih _) (Camera (V3 px py pz) rot) Settings {..} = do
  let (w, h)          = dimensions
      (nodes, tris)   = flattenBIH (tree bih)
      Bounds (V3 lx ly lz) (V3 hx hy hz) = bounds bih
  -- marshal: nodes (16 B each), triangles (40 B each, material index), materials (32 B each), camera (48 B)
  allocaBytes (16 * length nodes) $ \pNodes ->
   allocaBytes (40 * length tris)  $ \pTris  ->
   allocaBytes (32 * length tris)  $ \pMats  ->     -- one material per triangle is the simplest correct marshalling
   allocaBytes 48                  $ \pCam   ->
   allocaBytes 72                  $ \pScene -> do
    sequence_ [ do pokeByteOff pNodes (16*i) k; pokeByteOff pNodes (16*i+4) a
                   pokeByteOff pNodes (16*i+8) b; pokeByteOff pNodes (16*i+12) l
              | (i, (k, a, b, l)) <- zip [0..] nodes ]
    sequence_ [ do pokeV3 pTris (40*i) a; pokeV3 pTris (40*i+12) b; pokeV3 pTris (40*i+24) c
                   pokeByteOff pTris (40*i+36) (fromIntegral i :: Int32)
                   pokeByteOff pMats (32*i) ref; pokeV3 pMats (32*i+4) sc
                   pokeByteOff pMats (32*i+16) em; pokeV3 pMats (32*i+20) ec
              | (i, Triangle a b c (Mat ref sc em ec)) <- zip [0..] tris ]
    mapM_ (\(o, v) -> pokeByteOff pCam o (v :: Float)) (zip [0,4..] ([px, py, pz] ++ M.toList rot))
    mapM_ (\(o, v) -> pokeByteOff pScene o (v :: Float)) (zip [0,4..] [lx, ly, lz, hx, hy, hz])
    pokeByteOff pScene 24 pNodes; pokeByteOff pScene 32 (fromIntegral (length nodes) :: Int32)
    pokeByteOff pScene 40 pTris;  pokeByteOff pScene 48 (fromIntegral (length tris)  :: Int32)
    pokeByteOff pScene 56 pMats;  pokeByteOff pScene 64 (fromIntegral (length tris)  :: Int32)
    pokeByteOff pScene 68 (0 :: Int32)                                     -- height: 0 = let the library compute it
    out <- VSM.new (w * h * 3)
    rc  <- VSM.unsafeWith out $ \p ->
             c_sq_render_rgb8 (castPtr pScene) (castPtr pCam) (fromIntegral samples)
                              (fromIntegral w) (fromIntegral h) (if cast then 1 else 0) p
    if rc /= 0 then c_sq_last_error >>= peekCString >>= ioError . userError
    else do
      px <- VS.unsafeFreeze out
      -- w ROWS x h COLUMNS, 3 bytes per pixel: the same `Array S Ix2 (Pixel RGB Word8)` as src/Lib.hs:74
      let img = A.unsafeFromStorableVector Par (w :. h) (VS.unsafeCast px) :: Array S Ix2 (Pixel RGB Word8)
      writeImage (unSavePath savePath) img
  where pokeV3 p o (V3 x y z) = pokeByteOff p o x >> pokeByteOff p (o+4) y >> pokeByteOff p (o+8) z
