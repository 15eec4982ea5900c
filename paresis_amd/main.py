#!/usr/bin/env python3
"""XML-driven entry point (mirror of CodePython/main.py:20-115).

    python -m paresis_amd.main [--experiment NAME] [--type RayT|Fresnel] [--oversampling N] [--points N]
                               [--out DIR] [--format .tif|.edf|.npy] [--xml DIR] [--no-noise] [--seed S] [--backend nccl|gloo]

With torchrun (one process per GPU) the membrane positions are strided over the ranks and the detector images are
gathered on rank 0 over RCCL (paresis_amd/dist.py); results do not depend on the number of GPUs because every position
has its own seed.
"""
import argparse
import datetime
import os
import time

import numpy as np


def run(exp_dict, save=True, saving_format=".tif", backend=None):
    """main.py:58-115.  Returns on rank 0 {position: (Sample, Reference[, Propag, White, ...])} with host tensors."""
    from . import dist
    from .Experiment import Experiment
    from .InputOutput.pagailleIO import save_image

    time0 = time.time()
    rank, world = dist.init(backend)
    # one experiment ID for all ranks (the reference takes the wall clock, main.py:30; ranks would disagree by a second)
    exp_dict['expID'] = dist.broadcast_object(datetime.datetime.now().strftime("%Y%m%d-%H%M%S"), rank, world)
    exp_dict.setdefault('deferMeanEnergy', True)       # no host synchronisation per position (resolved before the dump)
    # this loop only reads the stacks a position returns (packs, copies to the host, saves): the all-zero Propag / White of the
    # positions after the first may be ONE stack of the experiment (Experiment._begin)
    exp_dict.setdefault('sharedZeroStacks', True)
    # exp_dict['reproducible'] (default True; --float-atomics turns it off): the ray-tracing chain's far rays go through the
    # order-independent replay (the Experiment class sets psx_set_deterministic around its chain and restores the caller's
    # mode): every image is then the same bits on 1 GPU and on 8; the Fresnel chain has no float atomics and is reproducible
    # as it is
    print("\n\nINITIALIZING EXPERIMENT PARAMETERS AND GEOMETRIES")
    experiment = Experiment(exp_dict)
    sim = exp_dict['simulation_type']
    root = exp_dict['filepath'] + ('Fresnel_' if sim == "Fresnel" else 'RayTracing_') + str(exp_dict['expID']) + '/'
    if save:
        if rank == 0:
            os.makedirs(root + 'membraneThickness/', exist_ok=True)                       # main.py:84-85
        dist.barrier()
    # everything allocated so far lives as long as the run: keep it out of the cyclic garbage collector's generations, or a
    # full collection (tens of ms with the GPU idle) lands in one of the first positions
    import gc
    gc.collect()
    gc.freeze()
    print("\nImages calculation")
    # the detector stacks go to rank 0 round by round while the next positions are computed (dist.PositionGatherer); the
    # Fresnel plan then hands out its line groups through a queue, so that the transfer's copy kernels cost it a few per cent
    dims = experiment.myDetector.det_param["myDimensions"]
    # Without shot noise the images are not photon counts: the packed 16-bit rounds would all be flagged and the gather
    # repeated in float32 at the end -- one float32 gather at the end from the start, then.  The overlapped form also needs
    # its buffers on every rank: rank 0 alone holds the receive buckets, so whether they could be allocated is decided
    # TOGETHER (a rank that fell back on its own would issue different collectives from the others and hang them).
    overlapped = world > 1 and bool(exp_dict.get('noise', True))
    gatherer = None
    if overlapped:
        gatherer = dist.PositionGatherer(exp_dict['nbExpPoints'], rank, world, to_host=True,
                                         shape=(experiment._close_bins(), int(dims[0]), int(dims[1])))
        if not dist.agree_on_overlap(gatherer):
            overlapped, gatherer = False, None
    if overlapped and sim == "Fresnel":
        experiment._plan().work_queue(True)
    experiment.reserve_outputs(len(dist.my_positions(exp_dict['nbExpPoints'], rank, world)) + 1)   # no hipMalloc inside the loop
    results = {}
    # the loop and the final gather share ONE failure path: PositionGatherer.add() issues collectives too, and a DistError
    # raised there must not unwind through the process group's teardown (its contract: leave with os._exit)
    try:
        for pointNum in dist.my_positions(exp_dict['nbExpPoints'], rank, world):
            experiment.myMembrane.myGeometry = []
            experiment.myMembrane.getMyGeometry(experiment.exp_dict['studyDimensions'], experiment.myMembrane.membranePixelSize,
                                                experiment.exp_dict['overSampling'], pointNum, exp_dict['nbExpPoints'])   # main.py:64-65
            print("\nCalculations point", pointNum)
            out = experiment.computeSampleAndReferenceImages(pointNum)
            if gatherer is not None:
                gatherer.add(pointNum, out)
            else:
                results[pointNum] = out
            if save and exp_dict.get('saveMembrane', True):
                # main.py:98: every rank writes the membrane maps of its own positions (one node, one file system: no gather)
                save_image(experiment.myMembrane.myGeometry[0], root + 'membraneThickness/' + exp_dict['experimentName'] +
                           '_sampling' + str(exp_dict['overSampling']) + '_' + str(pointNum) + saving_format)
        if gatherer is not None:
            gathered = gatherer.finish()
        else:
            gathered = dist.gather_positions(results, exp_dict['nbExpPoints'], rank, world, to_host=True,
                                             pack=bool(exp_dict.get('noise', True)))
    except dist.DistError as exc:             # the ranks are out of step: no further collective can be trusted
        import sys
        import traceback
        traceback.print_exc()
        sys.stderr.write("paresis_amd.main: %s -- leaving\n" % exc)
        sys.stderr.flush()
        os._exit(6)
    finally:
        gc.unfreeze()                         # run() is also an API: leave the collector as it was found
    experiment.resolve_mean_energy()
    if rank == 0 and save:
        os.makedirs(root, exist_ok=True)
        thresholds = [experiment.mySource.mySpectrum[0][0]] + list(experiment.myDetector.det_param['myBinsThersholds'])
        nbin = gathered[0][0].shape[0]
        paths = []
        for ibin in range(nbin):                                         # main.py:84-96
            p = root if nbin == 1 else f'{root}{"%2.2d" % thresholds[ibin]}_{"%2.2d" % thresholds[ibin + 1]}kev/'
            for sub in ("ref/", "sample/", "propag/"):
                os.makedirs(p + sub, exist_ok=True)
            paths.append(p)
        if sim == "RayT":
            # main.py:100-101 writes Df after every position to the SAME file: what stays is the last position's, which is
            # all zeros unless that position is 0 (darkFieldPropag only builds up there, EXP:491)
            N = tuple(int(v) for v in experiment.exp_dict['studyDimensions'])
            last = exp_dict['nbExpPoints'] - 1
            df = gathered[0][6] if last == 0 and len(gathered[0]) > 6 else np.zeros(N, dtype=np.float32)
            save_image(df, root + "DF" + saving_format)
        for pointNum in sorted(gathered):
            S, R = gathered[pointNum][0], gathered[pointNum][1]
            txt = '%2.2d' % pointNum
            for ibin in range(nbin):
                save_image(S[ibin], paths[ibin] + 'sample/sampleImage_' + str(exp_dict['expID']) + '_' + txt + saving_format)
                save_image(R[ibin], paths[ibin] + 'ref/ReferenceImage_' + str(exp_dict['expID']) + '_' + txt + saving_format)
                if pointNum == 0 and len(gathered[0]) > 3:
                    save_image(gathered[0][2][ibin], paths[ibin] + 'propag/PropagImage_' + str(exp_dict['expID']) + '_' + saving_format)
                    save_image(gathered[0][3][ibin], paths[ibin] + 'White_' + str(exp_dict['expID']) + '_' + saving_format)
        experiment.saveAllParameters(time0, exp_dict)
    dist.finish()
    print("\nfini")
    return gathered


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--experiment", default="Fil_Nylon_ID17")
    ap.add_argument("--type", default="RayT", choices=["RayT", "Fresnel"])
    ap.add_argument("--oversampling", type=int, default=2)
    ap.add_argument("--points", type=int, default=1)
    ap.add_argument("--out", default="Results/")
    ap.add_argument("--format", default=".tif")
    ap.add_argument("--xml", default=None)
    ap.add_argument("--no-noise", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--reproducible", action="store_true", help="accepted and ignored: the default since round 5")
    ap.add_argument("--float-atomics", action="store_true",
                    help="ray tracing: far rays summed with float atomics in arrival order (a few per cent faster; the last bit of "
                         "an image, and with it a Poisson draw, may differ between runs and GPU counts)")
    ap.add_argument("--backend", default=None, choices=[None, "nccl", "gloo"],
                    help="torch.distributed backend under torchrun (default: nccl = RCCL on a GPU node; gloo rehearses several "
                         "ranks on one GPU)")
    a = ap.parse_args(argv)
    exp_dict = {'experimentName': a.experiment, 'filepath': a.out if a.out.endswith('/') else a.out + '/',
                'overSampling': a.oversampling, 'nbExpPoints': a.points, 'simulation_type': a.type,
                'noise': not a.no_noise, 'seed': a.seed, 'reproducible': not a.float_atomics}
    if a.xml:
        exp_dict['xmlDir'] = a.xml
    os.makedirs(exp_dict['filepath'], exist_ok=True)
    run(exp_dict, save=True, saving_format=a.format, backend=a.backend)


if __name__ == "__main__":
    main()
