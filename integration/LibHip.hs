-- integration/LibHip.hs -- the Haskell side of the drop-in: what a maintainer of rrruko/squigly-trace adds as
-- src/LibHip.hs to run Lib.render's parallel fan-out (src/Lib.hs:73-74) on an MI355X through libsquigly_hip.so.
--
-- Shipped as source only: neither the build container nor the GPU box of this repository has GHC / stack / cabal
-- (DESIGN.md section 3), so it has never been compiled.  It was checked BY INSPECTION, name by name, against the
-- export lists and declarations it depends on (round 3; the list is at the end of this header).  The same C entry
-- points are exercised through ctypes by every GPU test (squigly-trace_amd/_native.py).
--
-- The complete change to the reference is integration/reference.patch (three hunks; INTEGRATION.md section 2):
--   src/BIH.hs:1-13        export Tree(..), BIHNode(..), BIHTree   (flattenBIH below pattern-matches on them)
--   squigly-trace.cabal    exposed-modules: + LibHip ; extra-libraries: squigly_hip
--   app/Main.hs:43         renderHip scene cam settings            (instead of render scene cam settings)
--
-- Names used, and where they come from (all paths relative to the reference root):
--   BIH(..) [bounds, tree], Scene(..)                 src/BIH.hs:2-3 (Scene re-exported from Geometry)
--   Tree(..) [Leaf, Branch], BIHNode(..) [BIHN], BIHTree   src/BIH.hs:26,37-40 -- NOT exported today: hunk 1 of the patch
--   Axis(..), Bounds(..), Camera(..), Triangle(..)    src/Geometry.hs:9-16,41,49-56,153
--   Material(..) [Mat], RGB = V3                      src/Color.hs:8-9,32,78-83
--   V3(..)                                            src/V3.hs:3-5 (module exports everything)
--   Settings(..), SavePath(..) [unSavePath]           src/Lib.hs:15-24,46-47,55-64
--   Settings {..}                                     needs RecordWildCards, as src/Lib.hs:13,69 does
--   Array, Comp(Par), D, S, Ix2((:.)), computeAs, makeArray, writeImage, Pixel, RGB, PixelRGB
--                                                     exactly the imports and uses of src/Lib.hs:33-40,66,70-75,104
{-# LANGUAGE ForeignFunctionInterface #-}
{-# LANGUAGE RecordWildCards          #-}
module LibHip (renderHip) where

import           BIH                          (BIH (..), BIHNode (..), BIHTree, Tree (..))
import           Color                        (Material (..))
import           Geometry                     (Axis (..), Bounds (..), Camera (..), Scene (..), Triangle (..))
import           Lib                          (SavePath (..), Settings (..))
import           V3                           (V3 (..))

import           Data.Int                     (Int32)
import           Data.Massiv.Array            (Array, Comp (..), D (..), Ix2 (..), S (..), computeAs, makeArray)
import           Data.Massiv.Array.IO         (writeImage)
import qualified Data.Matrix                  as Mx
import qualified Data.Vector                  as V
import qualified Data.Vector.Storable         as VS
import qualified Data.Vector.Storable.Mutable as VSM
import           Data.Word                    (Word8)
import           Foreign.C.String             (CString, peekCString)
import           Foreign.C.Types              (CInt (..))
import           Foreign.Marshal.Alloc        (allocaBytes)
import           Foreign.Ptr                  (Ptr, castPtr)
import           Foreign.Storable             (pokeByteOff)
import qualified Graphics.ColorSpace          as M

-- The C layouts (include/squigly_hip.h, x86-64):
--   typedef struct { int32_t kind; float lmax, rmin; int32_t link; } sq_node;     16 B, pre-order
--   typedef struct { float v0[3], v1[3], v2[3]; int32_t mat; } sq_tri;            40 B, BIH.flatten order
--   typedef struct { float reflective, surf[3], emissive, emit[3]; } sq_material; 32 B
--   typedef struct { float pos[3]; float rot[9]; } sq_camera;                     48 B
--   typedef struct { sq_bounds root;            offset  0 (24 B)
--                    const sq_node* nodes;      offset 24 ; int32_t n_nodes;  offset 32
--                    const sq_tri* tris;        offset 40 ; int32_t n_tris;   offset 48
--                    const sq_material* mats;   offset 56 ; int32_t n_mats;   offset 64
--                    int32_t height; } sq_scene;  offset 68 ; sizeof = 72
data SqScene   -- opaque to Haskell: filled with pokeByteOff below
data SqCamera

-- `safe`: the call runs for tens of milliseconds to seconds and must not block the RTS capability.
foreign import ccall safe "squigly_hip.h sq_render_rgb8"
  c_sq_render_rgb8 :: Ptr SqScene -> Ptr SqCamera -> Int32 -> Int32 -> Int32 -> Int32 -> Ptr Word8 -> IO CInt
foreign import ccall unsafe "squigly_hip.h sq_last_error"
  c_sq_last_error :: IO CString

type Pixel = M.Pixel M.RGB Word8            -- as src/Lib.hs:66

-- One sq_node: (kind, lmax, rmin, link).  kind & 3 = 0/1/2 for a Branch on X/Y/Z, 3 for a Leaf whose triangle count
-- sits in kind >> 2; link = index of the right child (the left child is the next node) or of the leaf's first triangle.
type NodeRec = (Int32, Float, Float, Int32)

-- Flatten `Tree BIHNode (Vector Triangle)` (src/BIH.hs:26,37-40) to pre-order sq_node records with the triangles in
-- BIH.flatten order (src/BIH.hs:50-52).  Sizes are returned with the lists so that nothing is re-counted.
flattenBIH :: BIHTree -> ([NodeRec], [Triangle])
flattenBIH t = let (ns, _, ts, _) = go t 0 0 in (ns, ts)
  where
    -- go subtree (index of its root node) (index of its first triangle) = (nodes, #nodes, triangles, #triangles)
    go :: BIHTree -> Int -> Int -> ([NodeRec], Int, [Triangle], Int)
    go (Leaf ts) _ firstTri =
      let n = V.length ts
      in  ([(3 + 4 * fromIntegral n, 0, 0, fromIntegral firstTri)], 1, V.toList ts, n)
    go (Branch (BIHN ax lmax rmin) l r) me firstTri =
      let (ln, lc, lt, ltc) = go l (me + 1) firstTri
          rightIx           = me + 1 + lc
          (rn, rc, rt, rtc) = go r rightIx (firstTri + ltc)
      in  ((axisCode ax, lmax, rmin, fromIntegral rightIx) : ln ++ rn, 1 + lc + rc, lt ++ rt, ltc + rtc)
    axisCode :: Axis -> Int32
    axisCode X = 0
    axisCode Y = 1
    axisCode Z = 2

-- | Drop-in for Lib.render (src/Lib.hs:68-75): same arguments, same PNG.
renderHip :: Scene BIH -> Camera -> Settings -> IO ()
renderHip (Scene bih _) (Camera (V3 cx cy cz) rot) Settings {..} = do
  let (w, h)        = dimensions
      (nodes, tris) = flattenBIH (tree bih)
      nNodes        = length nodes
      nTris         = length tris
      Bounds (V3 lx ly lz) (V3 hx hy hz) = bounds bih
  -- marshal: nodes (16 B each), triangles (40 B each), materials (32 B each), camera (48 B), scene header (72 B)
  allocaBytes (16 * max 1 nNodes) $ \pNodes ->
   allocaBytes (40 * max 1 nTris) $ \pTris  ->
   allocaBytes (32 * max 1 nTris) $ \pMats  ->      -- one material per triangle is the simplest correct marshalling
   allocaBytes 48                 $ \pCam   ->
   allocaBytes 72                 $ \pScene -> do
    sequence_ [ do pokeByteOff pNodes (16*i) k;     pokeByteOff pNodes (16*i+4) a
                   pokeByteOff pNodes (16*i+8) b;   pokeByteOff pNodes (16*i+12) l
              | (i, (k, a, b, l)) <- zip [0..] nodes ]
    sequence_ [ do pokeV3 pTris (40*i) a; pokeV3 pTris (40*i+12) b; pokeV3 pTris (40*i+24) c
                   pokeByteOff pTris (40*i+36) (fromIntegral i :: Int32)
                   pokeByteOff pMats (32*i) ref;    pokeV3 pMats (32*i+4) sc
                   pokeByteOff pMats (32*i+16) em;  pokeV3 pMats (32*i+20) ec
              | (i, Triangle a b c (Mat ref sc em ec)) <- zip [0..] tris ]
    -- Data.Matrix.toList is row-major; `rot` is the 3x3 product of src/Geometry.hs:90-102
    mapM_ (\(o, v) -> pokeByteOff pCam o (v :: Float))   (zip [0,4..] ([cx, cy, cz] ++ Mx.toList rot))
    mapM_ (\(o, v) -> pokeByteOff pScene o (v :: Float)) (zip [0,4..] [lx, ly, lz, hx, hy, hz])
    pokeByteOff pScene 24 (pNodes :: Ptr ()); pokeByteOff pScene 32 (fromIntegral nNodes :: Int32)
    pokeByteOff pScene 40 (pTris  :: Ptr ()); pokeByteOff pScene 48 (fromIntegral nTris  :: Int32)
    pokeByteOff pScene 56 (pMats  :: Ptr ()); pokeByteOff pScene 64 (fromIntegral nTris  :: Int32)
    pokeByteOff pScene 68 (0 :: Int32)                                     -- height: 0 = let the library compute it
    out <- VSM.new (w * h * 3) :: IO (VSM.IOVector Word8)
    rc  <- VSM.unsafeWith out $ \p ->
             c_sq_render_rgb8 (castPtr pScene) (castPtr pCam) (fromIntegral samples)
                              (fromIntegral w) (fromIntegral h) (if cast then 1 else 0) p
    if rc /= 0
      then c_sq_last_error >>= peekCString >>= ioError . userError
      else do
        bytes <- VS.unsafeFreeze out
        -- w ROWS x h COLUMNS, 3 bytes per pixel (src/Lib.hs:70-71,80).  The image is rebuilt with the very calls
        -- src/Lib.hs:73-74 uses (makeArray / computeAs S), so that no other massiv-0.1.0.0 API is assumed.
        let at (y :. x) = let o = 3 * (y * h + x)
                          in  M.PixelRGB (bytes VS.! o) (bytes VS.! (o + 1)) (bytes VS.! (o + 2)) :: Pixel
            buf = makeArray Par (w :. h) at :: Array D Ix2 Pixel
            img = computeAs S buf
        img `seq` writeImage (unSavePath savePath) img
  where
    pokeV3 :: Ptr a -> Int -> V3 -> IO ()
    pokeV3 p o (V3 x y z) = pokeByteOff p o x >> pokeByteOff p (o + 4) y >> pokeByteOff p (o + 8) z
