"""Host-side mirror of the reference's mapping classes for the hot path.

Same names, argument meaning and error behaviour as
  Mapping.executeMapping(int threads, String reference, String input, String outputPrefix,
                         int mappingQualityFilter, String additionalOptions)   (Mapping.java:40-42)
  PARAsuiteMapping (+ setErrorProfileFilename / setIndelProfileFilename)      (PARAsuiteMapping.java:22-154)
  BWAMapping                                                                    (BWAMapping.java:20-128)
with the three `bwa` child processes replaced by calls into libparasuite_hip.so.  What the Java does
after `bwa samse` -- samtools view -bS / view -q, rm, mv (PARAsuiteMapping.java:102-152), later samtools
sort / index (Mapping.java:85-108) -- is offered as ONE library call, `samToFilteredBam` (SURVEY.md §8f rank 3,
`ps_sam_to_bam`: BAM, MAPQ filter, coordinate sort and .bai without samtools); `<prefix>.sam` is left in place
unless that method is called, and the MAPQ filter also exists on the SAM text.
"""
import os
import time

from . import capi


class ExternalCallErrorException(Exception):
    """mirror of mapping.ExternalCallErrorException: the failing 'command' is carried along"""

    def __init__(self, command):
        super().__init__(command)
        self.command = command

    def getMappingCommand(self):
        return self.command


class Mapping:
    """abstract mirror of mapping.Mapping (Mapping.java:26-206)"""

    def __init__(self):
        self._t0 = None

    def executeMapping(self, threads, reference, input, outputPrefix, mappingQualityFilter, additionalOptions):
        raise NotImplementedError

    def setTimeStart(self):
        self._t0 = time.time()

    def calculatePassedTime(self):
        return int(time.time() - self._t0)          # whole seconds, as Mapping.java:203-206

    def _call(self, what, fn, *args):
        """Mapping.executeCommand's contract (Mapping.java:151-198): non-zero status => abort with the command"""
        try:
            fn(*args)
        except capi.PsError as e:
            raise ExternalCallErrorException("%s: %s" % (what, e))

    def samToFilteredBam(self, outputPrefix, mappingQualityFilter, sortByCoordinateAndIndex=False, threads=8):
        """<prefix>.sam -> <prefix>.bam holding the reads with MAPQ >= filter (PARAsuiteMapping.java:102-152), optionally
        coordinate-sorted with <prefix>.bam.bai (Mapping.sortByCoordinateAndIndex, Mapping.java:85-108); removes the SAM"""
        self._call("samtools view -bS | view -q %d%s" % (mappingQualityFilter, " | sort | index" if sortByCoordinateAndIndex else ""),
                   capi.ps_sam_to_bam, outputPrefix + ".sam", outputPrefix + ".bam", mappingQualityFilter,
                   sortByCoordinateAndIndex, sortByCoordinateAndIndex, threads)
        os.remove(outputPrefix + ".sam")

    def _map_to_bam(self, what, threads, mm, ep, ip, reference, input, outputPrefix, mappingQualityFilter, sortByCoordinateAndIndex):
        """the mapping call and samToFilteredBam fused (`ps_map_to_bam`): <prefix>.bam straight from the alignment records, the 2 GB of
        <prefix>.sam per 10 M reads are never written; BGZF blocks are compressed while later pieces of the input are searched"""
        if not os.path.exists(reference + ".bwt"):
            self._call("bwa index " + reference, capi.ps_index, reference)
        self.setTimeStart()
        self._call(what + " | samtools view -bS | view -q %d%s" % (mappingQualityFilter, " | sort | index" if sortByCoordinateAndIndex else ""),
                   capi.ps_map_to_bam, threads, mm, ep, ip, reference, input, outputPrefix + ".bam", mappingQualityFilter,
                   sortByCoordinateAndIndex, sortByCoordinateAndIndex)
        self.seconds = self.calculatePassedTime()

    @staticmethod
    def filter_sam_mapq(sam_in, sam_out, min_mapq):
        """what `samtools view -q Q` keeps (PARAsuiteMapping.java:121-133), on SAM text"""
        with open(sam_in) as fi, open(sam_out, "w") as fo:
            for line in fi:
                if line.startswith("@") or int(line.split("\t", 5)[4]) >= min_mapq:
                    fo.write(line)


class BWAMapping(Mapping):
    """first pass with stock penalties: `bwa aln -t T -n <mm>` + `bwa samse` (BWAMapping.java:51-75)"""

    def executeMapping(self, threads, reference, input, outputPrefix, mappingQualityFilter, additionalOptions):
        if not os.path.exists(reference + ".bwt"):                      # BWAMapping.java:35-45
            self._call("bwa index " + reference, capi.ps_index, reference)
        self.setTimeStart()
        self._call("bwa aln -t %d -n %s %s %s | bwa samse" % (threads, additionalOptions, reference, input),
                   capi.ps_map, threads, additionalOptions, None, None, reference, input, outputPrefix + ".sam")
        self.seconds = self.calculatePassedTime()

    def executeMappingToBam(self, threads, reference, input, outputPrefix, mappingQualityFilter, additionalOptions, sortByCoordinateAndIndex=False):
        """executeMapping + samToFilteredBam in one library call (BWAMapping.java:51-128 end to end), no SAM file"""
        self._map_to_bam("bwa aln -t %d -n %s %s %s | bwa samse" % (threads, additionalOptions, reference, input), threads, additionalOptions,
                         None, None, reference, input, outputPrefix, mappingQualityFilter, sortByCoordinateAndIndex)

    def executeMappingWithProfile(self, threads, reference, input, outputPrefix, mappingQualityFilter, additionalOptions, maxReadLength):
        """the first pass of a --refine run and the ErrorProfiling step behind it (Main.java:288-334) as ONE call: the profile of
        the alignments with MAPQ >= mappingQualityFilter is counted from memory while the SAM is written (`ps_map_profiled`);
        returns the two file names Main hands to PARAsuiteMapping (<outputPrefix>.bam.errorprofile / .indelprofile)"""
        if not os.path.exists(reference + ".bwt"):
            self._call("bwa index " + reference, capi.ps_index, reference)
        self.setTimeStart()
        prefix = outputPrefix + ".bam"                                    # Main.java:335-338: genomicMappingFileName + ".errorprofile"
        self._call("bwa aln -t %d -n %s %s %s | bwa samse | ErrorProfiling" % (threads, additionalOptions, reference, input),
                   capi.ps_map_profiled, threads, additionalOptions, None, None, reference, input, outputPrefix + ".sam",
                   mappingQualityFilter, maxReadLength, prefix)
        self.seconds = self.calculatePassedTime()
        return prefix + ".errorprofile", prefix + ".indelprofile"


class PARAsuiteMapping(Mapping):
    """refine pass with the error profile: `bwa parasuite -t T -X mm -p EP -g IP` + `bwa samse`
    (PARAsuiteMapping.java:63-92)"""

    def __init__(self):
        super().__init__()
        self.errorProfileFilename = None
        self.indelProfileFilename = None

    def setErrorProfileFilename(self, errorProfileFilename):
        self.errorProfileFilename = errorProfileFilename

    def setIndelProfileFilename(self, indelProfileFilename):
        self.indelProfileFilename = indelProfileFilename

    def executeMapping(self, threads, reference, input, outputPrefix, mappingQualityFilter, additionalOptions):
        if not os.path.exists(reference + ".bwt"):                      # PARAsuiteMapping.java:45-55
            self._call("bwa index " + reference, capi.ps_index, reference)
        self.setTimeStart()
        if not self.errorProfileFilename:
            raise ExternalCallErrorException("bwa parasuite: no error profile set (-p)")
        self._call("bwa parasuite -t %d -X %s -p %s -g %s %s %s | bwa samse" %
                   (threads, additionalOptions, self.errorProfileFilename, self.indelProfileFilename, reference, input),
                   capi.ps_map, threads, additionalOptions, self.errorProfileFilename, self.indelProfileFilename,
                   reference, input, outputPrefix + ".sam")
        self.seconds = self.calculatePassedTime()

    def executeMappingToBam(self, threads, reference, input, outputPrefix, mappingQualityFilter, additionalOptions, sortByCoordinateAndIndex=False):
        """executeMapping + samToFilteredBam in one library call (PARAsuiteMapping.java:63-152 end to end), no SAM file"""
        if not self.errorProfileFilename:
            raise ExternalCallErrorException("bwa parasuite: no error profile set (-p)")
        self._map_to_bam("bwa parasuite -t %d -X %s -p %s -g %s %s %s | bwa samse" %
                         (threads, additionalOptions, self.errorProfileFilename, self.indelProfileFilename, reference, input), threads, additionalOptions,
                         self.errorProfileFilename, self.indelProfileFilename, reference, input, outputPrefix, mappingQualityFilter, sortByCoordinateAndIndex)


class ErrorProfiling:
    """mirror of utils.errorprofile.ErrorProfiling (ErrorProfiling.java:57-88, 100-631) for what the mapping step uses of it:
    `new ErrorProfiling(mappingFileName, referenceFileName, maxReadLength).inferErrorProfile(false, false)` leaves
    <mappingFileName>.errorprofile and <mappingFileName>.indelprofile behind (Main.java:327-334).  The counting runs on the GPU
    (`ps_error_profile`); the other files the Java writes (.qualities, .indels, .vcf, plots) have no consumer in the path."""

    def __init__(self, mappingFileName, referenceFileName, maxReadLength):
        self.mappingFileName = mappingFileName
        self.referenceFileName = referenceFileName
        self.maxReadLength = maxReadLength

    def inferErrorProfile(self, isInferQualities=False, isShowErrorPlot=False):
        try:
            capi.ps_error_profile(self.mappingFileName, self.referenceFileName, self.maxReadLength, None)
        except capi.PsError as e:
            raise ExternalCallErrorException("ErrorProfiling %s: %s" % (self.mappingFileName, e))
        return self.mappingFileName + ".errorprofile", self.mappingFileName + ".indelprofile"
