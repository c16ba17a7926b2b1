-- integration/DumpGolden.hs -- turns "parity unpinned" into one command for anyone with a GHC build of the reference.
--
-- This repository checks its HIP kernels against a C restatement of the Haskell (oracle/sq_oracle.c), because no
-- Haskell toolchain exists where it is built.  Two things in that restatement cannot be checked offline: the word
-- order of tf-random's TFGen output (SURVEY.md App. B) and the host libm behind sin/cos/acos/atan.  This program,
-- compiled against the reference's OWN modules, writes the vectors that settle both:
--
--   ghc_tfgen_words.bin      4 seeds {0, 1, 2, 2^32+5} x the first 8 Word32 of `mkTFGen seed`   (128 bytes, LE)
--   ghc_avg_64x64_4spp.bin   the pre-tonemap `avg` of src/Lib.hs:88 for `-d 64,64 -s 4` on data/scene.obj,
--                            64 rows x 64 columns x RGB, Float32 LE (49152 bytes)
--
-- In a checkout of rrruko/squigly-trace (resolver lts-9.8, as its stack.yaml pins):
--     cp <this repo>/integration/DumpGolden.hs .
--     stack ghc -- -isrc -O1 DumpGolden.hs -o dump-golden
--     ./dump-golden <this repo>/tests/golden          # reads ./data/scene.obj, ./data/scene.sq, ./data/camera
-- then, in this repository:  python -m pytest tests/test_oracle.py -k ghc     (the test is skipped until the files exist)
--
-- `renderPixel` and `makeRay` are not exported by Lib (src/Lib.hs:15-24), so their few lines are repeated below,
-- each with its source line; everything that matters numerically -- raytrace, bounceRay, the RNG, intersectBIH,
-- makeBIH, the loaders, rotVert -- is the reference's own code.  Never compiled in this repository (no GHC here).
module Main (main) where

import           BIH                     (BIH, intersectBIH, makeBIH)
import           Geometry                (Camera (..), Ray (..), Scene (..), rotVert)
import           Lib                     (raytrace)
import           Obj                     (loadCamera, trisFromObj)
import           V3

import qualified Data.ByteString.Builder as B
import           Data.Monoid             ((<>))
import           Data.Word               (Word32)
import           System.Environment      (getArgs)
import           System.IO               (IOMode (WriteMode), withBinaryFile)
import           System.Random.TF        (TFGen, mkTFGen)
import           System.Random.TF.Gen    (next)

-- src/Lib.hs:107-114 (makeRay), with massiv's `w :. h` / `y :. x` written as plain Ints
makeRay :: Int -> Int -> Int -> Int -> Camera -> Ray
makeRay w h y x cam =
    let ww = fromIntegral w
        hh = fromIntegral h
        xoffs = (fromIntegral x - (ww / 2)) / ww
        yoffs = ((hh / 2) - fromIntegral y) / hh
        dir = V3 1 xoffs yoffs `rotVert` rotation cam
    in  Ray (position cam) dir

-- src/Lib.hs:79-88 (renderPixel) up to `avg`, the value rgbFloatToPixelRGB receives
pixelAvg :: Scene BIH -> Camera -> Int -> Int -> Int -> Int -> Int -> V3
pixelAvg scene cam sampleCount w h y x =
    let ray      = makeRay w h y x cam
        rix      = sampleCount * (x + y * w)
        rngs     = take sampleCount $ map mkTFGen [rix ..]
        outcomes = map (\r -> raytrace r scene ray 0) rngs
    in  (1 / fromIntegral sampleCount) *^ sum outcomes

firstWords :: Int -> TFGen -> [Word32]
firstWords 0 _ = []
firstWords n g = let (v, g') = next g in v : firstWords (n - 1) g'

main :: IO ()
main = do
    [outDir] <- getArgs
    obj  <- readFile "./data/scene.obj"
    tris <- trisFromObj False obj                       -- reads ./data/<mtllib>, src/Obj.hs:52
    cam  <- loadCamera "./data/camera"
    let scene  = Scene (makeBIH tris) intersectBIH      -- app/Main.hs:55-56
        (w, h, n) = (64, 64, 4)
        seeds  = [0, 1, 2, 2 ^ (32 :: Int) + 5] :: [Int]
        words' = concatMap (firstWords 8 . mkTFGen) seeds
        avgs   = [ pixelAvg scene cam n w h y x | y <- [0 .. w - 1], x <- [0 .. h - 1] ]   -- w ROWS, h COLUMNS (src/Lib.hs:70-71,80)
    withBinaryFile (outDir ++ "/ghc_tfgen_words.bin") WriteMode $ \hd ->
        B.hPutBuilder hd (mconcat (map B.word32LE words'))
    withBinaryFile (outDir ++ "/ghc_avg_64x64_4spp.bin") WriteMode $ \hd ->
        B.hPutBuilder hd (mconcat [ B.floatLE r <> B.floatLE g <> B.floatLE b | V3 r g b <- avgs ])
    putStrLn ("wrote ghc_tfgen_words.bin and ghc_avg_64x64_4spp.bin to " ++ outDir)
