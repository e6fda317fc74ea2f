{-# LANGUAGE DataKinds #-}
{-# LANGUAGE ViewPatterns #-}

-- | BASELINE.json configs[0] on the REFERENCE's own path (ad + hmatrix + hmatrix-gsl), timed: one double pendulum
--   (System 4 2), N x @stepHam 0.01@ from the reference example's initial state, and -- second figure -- the fixed-step
--   RK4 over 'hamEqs' that the GPU metric counts (RK4 phase-space steps per second, one host thread).
--
--   SURVEY.md section 8d, CPU-baseline step (1): "a Haskell driver that times the real stepHam when the toolchain is
--   present".  It cannot be built in the image this repository is developed in (no GHC, no GSL); @bench.py@ builds and runs
--   it when @ghc@, @cabal@ and @gsl-config@ are all on the PATH and reports the line below as
--   @cpu_baseline.reference_haskell.measured@.  Prints ONE JSON line.
--
--   usage: c1 [calls (default 1000)] [rk4 steps (default 20000)]
module Main (main) where

import           Control.Exception        (evaluate)
import           Data.List                (foldl', intercalate)
import           Data.Time.Clock          (diffUTCTime, getCurrentTime)
import qualified Data.Vector.Sized        as V
import           Numeric.Hamilton
import qualified Numeric.LinearAlgebra    as LA
import           Numeric.LinearAlgebra.Static
import           System.Environment       (getArgs)

-- The reference's double pendulum (app/Examples.hs:75-94), restated: two unit masses, rod 1 of length 1 from the pivot,
-- rod 2 of length 1/2; cartesian (x1, y1, x2, y2) with y measured from 1 below the pivot; U = 5 (m1 y1 + m2 y2).
pendulumPair :: System 4 2
pendulumPair =
  mkSystem' (vec4 1 1 1 1)
    (\q -> let a = V.index q 0
               b = V.index q 1
            in V.fromTuple (sin a, 1 - cos a, sin a + sin b / 2, 1 - cos a - cos b / 2))
    (\x -> 5 * (V.index x 1 + V.index x 3))

start :: Phase 2
start = toPhase pendulumPair (Cfg (vec2 (pi / 2) 0) (vec2 0 0))

-- classic RK4 over hamEqs: what `hamk_rk4_steps` runs per trajectory on the GPU
rk4 :: Double -> Phase 2 -> Phase 2
rk4 h y = y `plus` scale (h / 6) (k1 `plus` scale 2 k2 `plus` scale 2 k3 `plus` k4)
  where
    f p = let (dq, dp) = hamEqs pendulumPair p in Phs dq dp
    k1 = f y
    k2 = f (y `plus` scale (h / 2) k1)
    k3 = f (y `plus` scale (h / 2) k2)
    k4 = f (y `plus` scale h k3)
    plus (Phs a b) (Phs c d) = Phs (a + c) (b + d)
    scale c (Phs a b) = Phs (konst c * a) (konst c * b)

force :: Phase 2 -> IO (Phase 2)
force p@(Phs q m) = do
  _ <- evaluate (norm_2 q + norm_2 m)
  return p

list :: Phase 2 -> ([Double], [Double])
list (Phs q m) = (LA.toList (extract q), LA.toList (extract m))

main :: IO ()
main = do
  args <- getArgs
  let calls = case args of (a : _) -> read a; _ -> 1000 :: Int
      steps = case args of (_ : b : _) -> read b; _ -> 20000 :: Int
  _ <- force (iterate (stepHam 0.01 pendulumPair) start !! 20)          -- warm
  t0 <- getCurrentTime
  p1 <- force (foldl' (\p _ -> stepHam 0.01 pendulumPair p) start [1 .. calls])
  t1 <- getCurrentTime
  p2 <- force (foldl' (\p _ -> rk4 0.01 p) start [1 .. steps])
  t2 <- getCurrentTime
  let us = realToFrac (diffUTCTime t1 t0) * 1e6 / fromIntegral calls :: Double
      rate = fromIntegral steps / realToFrac (diffUTCTime t2 t1) :: Double
      (q1, m1) = list p1
      (q2, m2) = list p2
      arr xs = "[" ++ intercalate ", " (map show xs) ++ "]"
  putStrLn $ "{\"workload\": \"doublePendulum, 1 trajectory, " ++ show calls ++ " x stepHam 0.01 (BASELINE.json configs[0]), reference ad+hmatrix+hmatrix-gsl path\""
          ++ ", \"stepham_us_per_call\": " ++ show us
          ++ ", \"rk4_steps_per_s_one_thread\": " ++ show rate
          ++ ", \"rk4_steps\": " ++ show steps
          ++ ", \"q_after_stepham\": " ++ arr q1 ++ ", \"p_after_stepham\": " ++ arr m1
          ++ ", \"q_after_rk4\": " ++ arr q2 ++ ", \"p_after_rk4\": " ++ arr m2 ++ "}"
