{-# LANGUAGE DataKinds                 #-}
{-# LANGUAGE ExistentialQuantification #-}
{-# LANGUAGE RankNTypes                #-}
{-# LANGUAGE ScopedTypeVariables       #-}

-- | Golden emitter: what the REFERENCE ITSELF returns at the points of this repository's fixtures.
--
--   The reference (mstksg/hamilton) holds no test vectors for its equations-of-motion path and cannot be built in the image
--   this repository is developed in (no GHC, no GSL): every parity statement of the repository is therefore made against a
--   restatement ("parity unpinned").  This program is the way out for whoever has the toolchain.  Built against an UNPATCHED
--   checkout -- it uses the public API of @Numeric.Hamilton@ and nothing else -- it evaluates, for the six example systems of
--   @app/Examples.hs@ (restated below through that API; the executable's module cannot be imported) at the points of
--   @tests/golden/<system>.json@ (module "Points", generated):
--
--     underlyingPos, momenta / toPhase, velocities, keC, keP, pe, lagrangian, hamiltonian, hamEqs     (Hamilton.hs:174-387)
--     one @stepHam 0.01@ from every point                                                              (Hamilton.hs:390-402)
--     @evolveHam'@ on the grid [0, 0.01, 0.02, 0.05, 0.1] from point 0 (row 0 = the initial state)    (Hamilton.hs:409-462)
--
--   and prints ONE JSON document.  @tests/test_reference_haskell.py@ compares the C oracle and the HIP kernels with it when
--   @tests/golden/reference_haskell/emitted.json@ exists (and SKIPS, not passes, otherwise):
--
--   > cd bindings/haskell/golden && echo "packages: /path/to/hamilton" > cabal.project.local
--   > cabal run emit-golden > ../../../tests/golden/reference_haskell/emitted.json
--
--   Doubles are printed with 'show' (shortest decimal that reads back as the same Double).
module Main (main) where

import           Data.List                    (intercalate)
import qualified Data.Vector.Sized            as V
import           GHC.TypeLits                 (KnownNat)
import           Numeric.Hamilton
import qualified Numeric.LinearAlgebra        as LA
import           Numeric.LinearAlgebra.Static (R, extract, vector)
import           Points                       (points)

-- | One example: its fixture name and the system, sizes hidden.
data Example = forall m n. (KnownNat m, KnownNat n) => Example String (System m n)

-- ---------------------------------------------------------------------------------------------------------------------
-- The six systems of app/Examples.hs with its CLI defaults (Examples.hs:230-359), written against the public API.
-- ---------------------------------------------------------------------------------------------------------------------

-- | @logistic pos ht width@ (Examples.hs:601-605): a smooth step of height @ht@ at @pos@, 10 % .. 90 % over @width@.
wall :: Floating a => a -> a -> a -> a -> a
wall pos ht width x = ht / (1 + exp (negate (beta * (x - pos))))
  where beta = log (0.9 / (1 - 0.9)) / width

-- Examples.hs:61-73
onePendulum :: System 2 1
onePendulum =
  mkSystem' (vector [1, 1])
    (\q -> let t = V.index q 0 in V.fromTuple (sin t, 0.5 - cos t))
    (\x -> V.index x 1)

-- Examples.hs:75-94, m1 = m2 = 1
pendulumPair :: System 4 2
pendulumPair =
  mkSystem' (vector [1, 1, 1, 1])
    (\q -> let a = V.index q 0
               b = V.index q 1
            in V.fromTuple (sin a, 1 - cos a, sin a + sin b / 2, 1 - cos a - cos b / 2))
    (\x -> 5 * (1 * V.index x 1 + 1 * V.index x 3))

-- Examples.hs:96-116
box :: System 2 2
box =
  mkSystem (vector [1, 1])
    id
    (\q -> let x = V.index q 0
               y = V.index q 1
            in sum [ 2 * y
                   , 1 - wall (-1) 10 0.1 y
                   , wall 1 10 0.1 y
                   , 1 - wall (-2) 10 0.1 x
                   , wall 2 10 0.1 x ])

-- Examples.hs:118-142, m1 = 5, m2 = 0.5
orbit :: System 4 2
orbit =
  mkSystem (vector [m1, m1, m2, m2])
    (\q -> let r  = V.index q 0
               t  = V.index q 1
               r1 = r * realToFrac (negate (m2 / mT))
               r2 = r * realToFrac (m1 / mT)
            in V.fromTuple (r1 * cos t, r1 * sin t, r2 * cos t, r2 * sin t))
    (\q -> negate (realToFrac (m1 * m2) / V.index q 0))
  where
    m1, m2, mT :: Double
    m1 = 5
    m2 = 0.5
    mT = m1 + m2

-- Examples.hs:144-162, mB = 2, mW = 1, k = 10  (the weight's gravity term multiplies by mB, as the source does at :157)
hangingSpring :: System 3 3
hangingSpring =
  mkSystem (vector [mB, mW, mW])
    (\q -> let r = V.index q 0
               x = V.index q 1
               t = V.index q 2
            in V.fromTuple (r, r + (1 + x) * sin t, (1 + x) * negate (cos t)))
    (\q -> let r = V.index q 0
               x = V.index q 1
               t = V.index q 2
            in realToFrac k * x ** 2 / 2
                 + (1 - wall (-1.5) 25 0.1 r)
                 + wall 1.5 25 0.1 r
                 + realToFrac mB * ((1 + x) * negate (cos t)))
  where
    mB, mW, k :: Double
    mB = 2
    mW = 1
    k  = 10

-- Examples.hs:164-183 with the default control points of Examples.hs:350 and bezierCurve of :607-627
beadOnCurve :: System 2 1
beadOnCurve =
  mkSystem (vector [1, 1])
    (\q -> let t = V.index q 0 in V.fromTuple (curve fst t, curve snd t))
    (\q -> let t = V.index q 0 in (1 - wall 0 5 0.05 t) + wall 1 5 0.05 t)
  where
    ctrl :: [(Double, Double)]
    ctrl = [(-1, -1), (-2, 1), (0, 1), (1, -1), (2, 1)]
    deg :: Int
    deg = length ctrl - 1
    choose :: Int -> Int -> Int
    choose a b = product [1 .. a] `div` (product [1 .. a - b] * product [1 .. b])
    -- sum_i C(deg, i) (1 - t)^(deg - i) t^i P_i, folded from 0 in the order of the control points
    curve :: Fractional a => ((Double, Double) -> Double) -> a -> a
    curve pick t =
      foldl (+) 0
        [ realToFrac (pick p) * (fromIntegral (deg `choose` i) * (1 - t) ^ (deg - i) * t ^ i)
        | (i, p) <- zip [0 ..] ctrl ]

examples :: [Example]
examples =
  [ Example "pendulum" onePendulum
  , Example "doublePendulum" pendulumPair
  , Example "room" box
  , Example "twoBody" orbit
  , Example "spring" hangingSpring
  , Example "bezier" beadOnCurve
  ]

-- ---------------------------------------------------------------------------------------------------------------------
-- JSON by hand (no aeson: the program depends on what the reference depends on)
-- ---------------------------------------------------------------------------------------------------------------------
arr :: [String] -> String
arr xs = "[" ++ intercalate ", " xs ++ "]"

nums :: [Double] -> String
nums = arr . map num

num :: Double -> String
num x
  | isNaN x      = "null"
  | isInfinite x = "null"
  | otherwise    = show x

obj :: [(String, String)] -> String
obj kvs = "{" ++ intercalate ", " [show k ++ ": " ++ v | (k, v) <- kvs] ++ "}"

toL :: KnownNat k => R k -> [Double]
toL = LA.toList . extract

phaseJ :: KnownNat n => Phase n -> String
phaseJ (Phs q p) = obj [("q", nums (toL q)), ("p", nums (toL p))]

stepDt :: Double
stepDt = 0.01

grid :: [Double]
grid = [0, 0.01, 0.02, 0.05, 0.1]

emit :: Example -> String
emit (Example name (s :: System m n)) =
  obj [ ("system", show name)
      , ("points", arr (map point pts))
      , ("evolve", obj [ ("ts", nums grid)
                       , ("from_point", "0")
                       , ("states", arr (map phaseJ (evolveHam' s (phaseOf (head pts)) grid))) ])
      ]
  where
    pts = points name
    cfgOf :: ([Double], [Double]) -> Config n
    cfgOf (q, qd) = Cfg (vector q) (vector qd)
    phaseOf :: ([Double], [Double]) -> Phase n
    phaseOf = toPhase s . cfgOf
    point :: ([Double], [Double]) -> String
    point pt@(q, qd) =
      let c        = cfgOf pt
          ph       = toPhase s c
          (dq, dp) = hamEqs s ph
       in obj [ ("q", nums q)
              , ("qd", nums qd)
              , ("x", nums (toL (underlyingPos s (vector q :: R n))))
              , ("p", nums (toL (momenta s c)))
              , ("vel", nums (toL (velocities s ph)))
              , ("keC", num (keC s c))
              , ("keP", num (keP s ph))
              , ("pe", num (pe s (vector q :: R n)))
              , ("lagrangian", num (lagrangian s c))
              , ("hamiltonian", num (hamiltonian s ph))
              , ("dq", nums (toL dq))
              , ("dp", nums (toL dp))
              , ("stepHam_dt", num stepDt)
              , ("stepHam", phaseJ (stepHam stepDt s ph))
              ]

main :: IO ()
main =
  putStrLn $
    obj [ ("generator", show "bindings/haskell/golden/EmitGolden.hs over Numeric.Hamilton (mstksg/hamilton, unpatched)")
        , ("epsilon_note", show "stepHam / evolveHam' are the reference's: GSL RKF45 through hmatrix-gsl odeSolveV, h0 = dt/100, eps 1.49012e-08")
        , ("systems", arr (map emit examples))
        ]
